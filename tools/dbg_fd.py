import sys, json
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
name = "iiwa14"
robot = RobotModel.from_fixture(name); n = robot.n
for N in (1024, 2048, 4096, 8192, 16384, 32768):
    lib = load(name, max_timesteps=N)
    rng = np.random.default_rng(0)
    x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((N, n), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for label, fn in (("fd", lambda: lib.forward_dynamics_device(d_in, N, d_out, stream=st)), ("id", lambda: lib.inverse_dynamics_device(d_in, None, N, d_out, stream=st))):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): fn()
        e1.record(); torch.cuda.synchronize()
        res[label] = round(1e3 * e0.elapsed_time(e1) / 100, 2)
    print(N, res)
    lib.close()
