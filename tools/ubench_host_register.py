#!/usr/bin/env python3
"""What does pinning a caller's pageable buffer cost per call?  (decides how grid_*_host overlaps its copies: hipHostRegister per call vs staging through the
handle's pinned buffers)  usage: python tools/ubench_host_register.py"""
import json, time
import numpy as np, torch
rt = torch.cuda.cudart()
torch.cuda.init(); torch.zeros(1, device="cuda")
for mb in (1.4, 6.4, 25.6):
    n = int(mb * 1e6) // 4
    a = np.empty(n, np.float32); a[:] = 1.0
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
        t1 = time.perf_counter()
        rt.cudaHostUnregister(a.ctypes.data)
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    ts = np.array(ts) * 1e6
    d = torch.empty(n, dtype=torch.float32, device="cuda")
    h = torch.from_numpy(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): h.copy_(d); torch.cuda.synchronize()
    pageable_us = (time.perf_counter() - t0) / 10 * 1e6
    hp = torch.empty(n, dtype=torch.float32).pin_memory()
    t0 = time.perf_counter()
    for _ in range(10): hp.copy_(d, non_blocking=True); torch.cuda.synchronize()
    pinned_us = (time.perf_counter() - t0) / 10 * 1e6
    t0 = time.perf_counter()
    for _ in range(10): a[:] = hp.numpy()
    memcpy_us = (time.perf_counter() - t0) / 10 * 1e6
    print(json.dumps({"MB": mb, "register_us_median": float(np.median(ts[:, 0])), "unregister_us_median": float(np.median(ts[:, 1])), "rc": int(rc),
                      "d2h_pageable_us": pageable_us, "d2h_pinned_us": pinned_us, "host_memcpy_us": memcpy_us}))
