"""PCIe-inclusive rate of the host-buffer entry point (reference host-wrapper semantics: H2D, launch, D2H, synchronous).  Never the headline value."""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
name, N = "iiwa14", 16384
robot = RobotModel.from_fixture(name); n = robot.n
lib = load(name, max_timesteps=N)
rng = np.random.default_rng(0)
x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
for _ in range(5): lib.forward_dynamics_gradient_host(x)
t0 = time.perf_counter(); K = 50
for _ in range(K): lib.forward_dynamics_gradient_host(x)
dt = (time.perf_counter() - t0) / K
print(json.dumps({"entry": "grid_forward_dynamics_gradient_host", "batch": N, "us_per_call": round(1e6 * dt, 1), "solves_per_s": round(N / dt),
                  "bytes_moved": int(x.nbytes + N * 2 * n * n * 4), "effective_GBps": round((x.nbytes + N * 2 * n * n * 4) / dt / 1e9, 1)}))
