#!/usr/bin/env python3
"""Rate of the T = double instantiation of the hot path through the C ABI (grid_forward_dynamics_gradient_device_f64) next to the float one.
usage: python tools/bench_f64.py [robot:batch ...]"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
for spec in (sys.argv[1:] or ["iiwa14:16384", "hyq:4096", "atlas:16384"]):
    name, N = spec.split(":"); N = int(N)
    n = RobotModel.from_fixture(name).n
    lib = load(name, max_timesteps=N)
    rng = np.random.default_rng(0)
    x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))])
    st = torch.cuda.current_stream().cuda_stream
    res = {"robot": name, "batch": N}
    outs = {}
    for tag, dt, fn in (("f32", torch.float32, lib.forward_dynamics_gradient_device), ("f64", torch.float64, lib.forward_dynamics_gradient_device_f64)):
        d_in = torch.from_numpy(x).to(dt).cuda(); d_out = torch.empty((N, 2 * n * n), dtype=dt, device="cuda")
        t_end = time.perf_counter() + 0.1
        while time.perf_counter() < t_end:
            for _ in range(5): fn(d_in, N, d_out, stream=st)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 30
        e0.record()
        for _ in range(K): fn(d_in, N, d_out, stream=st)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / K
        res[tag + "_us_per_launch"] = round(us, 2); res[tag + "_solves_per_s"] = round(N / us * 1e6)
        outs[tag] = d_out.double().cpu().numpy()
    res["max_rel_diff_f32_vs_f64"] = float((np.abs(outs["f32"] - outs["f64"]).max(axis=1) / np.abs(outs["f64"]).max(axis=1)).max())
    print(json.dumps(res), flush=True)
    lib.close()
