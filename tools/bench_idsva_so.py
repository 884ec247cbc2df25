#!/usr/bin/env python3
"""Timing of the second-order inverse-dynamics derivative kernel (SURVEY.md section 8(f) rank 3 / BASELINE.json config 5).  usage: python tools/bench_idsva_so.py [robot] [batch ...]"""
import json, sys
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
name = sys.argv[1] if len(sys.argv) > 1 else "iiwa14"
batches = [int(a) for a in sys.argv[2:]] or [1024, 16384, 65536]
robot = RobotModel.from_fixture(name); n = robot.n
for N in batches:
    import os
    lib = load(name, max_timesteps=N, build_dir=os.environ.get("GRID_SWEEP_BUILD_DIR"))
    if os.environ.get("GRID_SWEEP_THREADS"):
        lib.set_launch_dims(0, int(os.environ["GRID_SWEEP_THREADS"]))
    rng = np.random.default_rng(0)
    x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
    d_in = torch.from_numpy(x).cuda(); d_qdd = torch.from_numpy(rng.uniform(-5, 5, (N, n)).astype(np.float32)).cuda()
    d_out = torch.empty((N, 4 * n ** 3), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    import time
    t_end = time.perf_counter() + 0.1  # clock warm
    while time.perf_counter() < t_end:
        for _ in range(3): lib.idsva_so_device(d_in, d_qdd, N, d_out, stream=st)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 30
    e0.record()
    for _ in range(K): lib.idsva_so_device(d_in, d_qdd, N, d_out, stream=st)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / K
    by = 4 * (3 * n + 4 * n ** 3)
    print(json.dumps({"robot": name, "kernel": "idsva_so (with qdd)", "batch": N, "us_per_launch": round(us, 2), "solves_per_s": round(N / us * 1e6),
                      "hbm_GBps_algorithmic": round(by * N / us / 1e3, 1), "hbm_frac_of_8TBps": round(by * N / us / 1e3 / 8000, 3)}))
    d_out2 = torch.empty((N, 4 * n ** 3), dtype=torch.float32, device="cuda")
    for _ in range(3): lib.fdsva_so_device(d_in, N, d_out2, stream=st)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(K): lib.fdsva_so_device(d_in, N, d_out2, stream=st)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / K
    print(json.dumps({"robot": name, "kernel": "fdsva_so", "batch": N, "us_per_launch": round(us, 2), "solves_per_s": round(N / us * 1e6),
                      "hbm_GBps_algorithmic": round(by * N / us / 1e3, 1), "hbm_frac_of_8TBps": round(by * N / us / 1e3 / 8000, 3)}))
    lib.close()
