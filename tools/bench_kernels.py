#!/usr/bin/env python3
"""Launch times of every device entry point of one robot library at one batch size (no oracle, product path only).
usage: python tools/bench_kernels.py <robot> <batch> [build-dir]"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
name, N = sys.argv[1], int(sys.argv[2])
bdir = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
n = RobotModel.from_fixture(name).n
lib = load(name, max_timesteps=N, build_dir=bdir)
rng = np.random.default_rng(0)
x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
st = torch.cuda.current_stream().cuda_stream
d_x = torch.from_numpy(x).cuda(); d_qdd = torch.from_numpy(rng.uniform(-5, 5, (N, n)).astype(np.float32)).cuda()
E = lambda c: torch.empty((N, c), dtype=torch.float32, device="cuda")
d_c, d_M, d_g, d_a = E(n), E(n * n), E(2 * n * n), E(n)
cases = [("inverse_dynamics", lambda: lib.inverse_dynamics_device(d_x, d_qdd, N, d_c, stream=st)),
         ("inverse_dynamics_gradient", lambda: lib.inverse_dynamics_gradient_device(d_x, d_qdd, N, d_g, stream=st)),
         ("direct_minv", lambda: lib.direct_minv_device(d_x, N, d_M, stream=st)),
         ("forward_dynamics", lambda: lib.forward_dynamics_device(d_x, N, d_a, stream=st)),
         ("aba", lambda: lib.aba_device(d_x, N, d_a, stream=st)),
         ("forward_dynamics_gradient", lambda: lib.forward_dynamics_gradient_device(d_x, N, d_g, stream=st))]
t_end = time.perf_counter() + 0.15
while time.perf_counter() < t_end:
    for _, fn in cases: fn()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for nm, fn in cases:
    res = []
    for rep in range(3):
        for _ in range(5): fn()
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(1e3 * e0.elapsed_time(e1) / 50)
    us = sorted(res)[1]
    print(json.dumps({"robot": name, "batch": N, "kernel": nm, "us_per_launch": round(us, 2), "solves_per_s": round(N / us * 1e6), "build": bdir or "shipped"}))
