#!/usr/bin/env python3
"""Timing of the (q, qd, qdd, Minv)-input overload of forward_dynamics_gradient_kernel (second overload) next to the u-input kernel:
qdd and M^-1 come from the library's own forward_dynamics / direct_minv kernels.  usage: python tools/bench_qdd_minv.py [robot batch]..."""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
cfgs = sys.argv[1:] or ["iiwa14", "16384", "hyq", "4096", "atlas", "16384"]
for name, N in zip(cfgs[0::2], (int(v) for v in cfgs[1::2])):
    n = RobotModel.from_fixture(name).n
    lib = load(name, max_timesteps=N)
    rng = np.random.default_rng(0)
    x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
    st = torch.cuda.current_stream().cuda_stream
    d_x = torch.from_numpy(x).cuda()
    d_qdd = torch.empty((N, n), dtype=torch.float32, device="cuda"); d_Minv = torch.empty((N, n * n), dtype=torch.float32, device="cuda")
    a, b = (torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda") for _ in range(2))
    lib.forward_dynamics_device(d_x, N, d_qdd, stream=st); lib.direct_minv_device(d_x, N, d_Minv, stream=st)
    run = {"forward_dynamics_gradient (u input)": lambda: lib.forward_dynamics_gradient_device(d_x, N, a, stream=st),
           "forward_dynamics_gradient (qdd, Minv input)": lambda: lib.forward_dynamics_gradient_qdd_minv_device(d_x, d_qdd, d_Minv, N, b, stream=st)}
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        for f in run.values(): f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for kern, f in run.items():
        ts = []
        for rep in range(5):
            e0.record()
            for _ in range(30): f()
            e1.record(); torch.cuda.synchronize(); ts.append(1e3 * e0.elapsed_time(e1) / 30)
        print(json.dumps({"robot": name, "batch": N, "kernel": kern, "us_per_launch": round(sorted(ts)[2], 2), "solves_per_s": round(N / sorted(ts)[2] * 1e6)}), flush=True)
    print(json.dumps({"robot": name, "batch": N, "max_rel_diff_between_the_two": float((a - b).abs().max() / a.abs().max())}), flush=True)
    lib.close()
