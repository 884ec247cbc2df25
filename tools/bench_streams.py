"""Aside (not the headline protocol): K steps issued round-robin on S HIP streams, each stream with its own output buffer,
so that consecutive 16384-solve launches overlap on the GPU (one launch only provides 2 waves per SIMD)."""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
name, N, K = "iiwa14", 16384, 400
robot = RobotModel.from_fixture(name); n = robot.n
lib = load(name, max_timesteps=N)
rng = np.random.default_rng(0)
x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
d_in = torch.from_numpy(x).cuda()
for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [torch.empty((N, 2*n*n), dtype=torch.float32, device="cuda") for _ in range(S)]
    for i in range(20): lib.forward_dynamics_gradient_device(d_in, N, outs[i % S], stream=streams[i % S].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): lib.forward_dynamics_gradient_device(d_in, N, outs[i % S], stream=streams[i % S].cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"streams": S, "us_per_step": round(1e6 * dt / K, 2), "solves_per_s": round(N * K / dt)}))
