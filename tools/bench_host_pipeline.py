#!/usr/bin/env python3
"""End-to-end rate of grid_forward_dynamics_gradient_host (H2D + kernel + D2H, synchronous) by buffer kind and chunk count.  usage: python tools/bench_host_pipeline.py [robot] [batch]"""
import ctypes, json, sys, time
sys.path.insert(0, ".")
import numpy as np
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
name = sys.argv[1] if len(sys.argv) > 1 else "iiwa14"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
n = RobotModel.from_fixture(name).n
lib = load(name, max_timesteps=N)
rng = np.random.default_rng(0)
x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
out = np.empty((N, 2 * n * n), np.float32)
xp, op = lib.pinned_empty(x.shape), lib.pinned_empty(out.shape)
xp[:] = x
def rate(fn, reps=50):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t0) / reps
    return {"us_per_call": round(dt * 1e6, 1), "solves_per_s": round(N / dt)}
print(json.dumps(dict(kind="pageable in + fresh pageable out", **rate(lambda: lib.forward_dynamics_gradient_host(x)))))
print(json.dumps(dict(kind="pageable in + reused pageable out", **rate(lambda: lib.forward_dynamics_gradient_host(x, out=out)))))
for ch in (1, 2, 3, 4, 8):
    lib._check(lib.lib.grid_set_host_chunks(lib.handle, ctypes.c_int(ch)))
    print(json.dumps(dict(kind="page-locked in + out, %d chunk(s)" % ch, **rate(lambda: lib.forward_dynamics_gradient_host(xp, out=op)))))
assert np.array_equal(op, lib.forward_dynamics_gradient_host(x))
