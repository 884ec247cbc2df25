#!/usr/bin/env python3
"""Does the launch time of one library depend on where the caching allocator happened to put the buffers?  usage: python tools/bench_alloc_modes.py <robot> <batch>"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
name, N = sys.argv[1], int(sys.argv[2])
n = RobotModel.from_fixture(name).n
lib = load(name, max_timesteps=N)
rng = np.random.default_rng(0)
x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
st = torch.cuda.current_stream().cuda_stream
keep = []
for trial in range(8):
    junk = torch.empty(int(1 + trial * 3.3e6), dtype=torch.uint8, device="cuda") if trial % 2 else None
    d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        for _ in range(20): lib.forward_dynamics_gradient_device(d_in, N, d_out, stream=st)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = []
    for rep in range(3):
        e0.record()
        for _ in range(50): lib.forward_dynamics_gradient_device(d_in, N, d_out, stream=st)
        e1.record(); torch.cuda.synchronize()
        res.append(round(1e3 * e0.elapsed_time(e1) / 50, 2))
    print(json.dumps({"robot": name, "trial": trial, "us": res, "out_ptr_mod_2M": d_out.data_ptr() % (2 << 20), "in_ptr_mod_2M": d_in.data_ptr() % (2 << 20), "out_ptr": hex(d_out.data_ptr())}))
    keep.append((junk, d_in, d_out))  # (never freed: every trial gets fresh addresses)
