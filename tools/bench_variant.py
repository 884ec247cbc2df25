import sys, json
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
name, N = sys.argv[1], int(sys.argv[2])
bdir = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
robot = RobotModel.from_fixture(name); n = robot.n
lib = load(name, max_timesteps=N, build_dir=bdir)
if threads: lib.set_launch_dims(0, threads)
rng = np.random.default_rng(0)
x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((N, 2*n*n), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
import time
t_end = time.perf_counter() + 0.15  # clock warm: 150 ms of back-to-back launches (a cold GPU ramps its clocks over milliseconds; without this the
while time.perf_counter() < t_end:  # short timed region below lands on a ramping clock and repeats of one build differ by 40 %)
    for _ in range(20): lib.forward_dynamics_gradient_device(d_in, N, d_out, stream=st)
    torch.cuda.synchronize()
for _ in range(10): lib.forward_dynamics_gradient_device(d_in, N, d_out, stream=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 50
e0.record()
for _ in range(K): lib.forward_dynamics_gradient_device(d_in, N, d_out, stream=st)
e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / K
print(json.dumps({"robot": name, "variant": bdir or "default", "threads": threads or lib.suggested_threads, "batch": N, "lanes_per_solve": lib.lanes_per_solve, "us_per_launch": round(us, 2), "solves_per_s": round(N / us * 1e6)}))
