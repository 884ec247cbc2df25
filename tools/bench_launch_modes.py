"""How K back-to-back 16384-solve passes are best issued (experiment behind bench.py's protocol notes).
Modes: one stream (the bench protocol); S streams round-robin with one output buffer per stream; a hipGraph of K kernel nodes (linear chain, captured
from one stream; and S parallel chains captured from S streams).  usage: python tools/bench_launch_modes.py [K]"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load

name, N = "iiwa14", 16384
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
robot = RobotModel.from_fixture(name)
n = robot.n
lib = load(name, max_timesteps=N)
rng = np.random.default_rng(0)
x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
d_in = torch.from_numpy(x).cuda()


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), float(np.min(ts))


# warm clocks
o = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
st = lib.prepare_forward_dynamics_gradient_device(d_in, N, o, stream=torch.cuda.current_stream().cuda_stream)
t_end = time.perf_counter() + 0.1
while time.perf_counter() < t_end:
    for _ in range(64):
        st()
    torch.cuda.synchronize()

for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda") for _ in range(S)]
    steps = [lib.prepare_forward_dynamics_gradient_device(d_in, N, outs[i], stream=streams[i].cuda_stream) for i in range(S)]

    def run():
        for i in range(K):
            steps[i % S]()
    med, mn = timed(run)
    print(json.dumps({"mode": "%d stream(s), prepared launches" % S, "K": K, "us_per_step_median": round(1e6 * med / K, 2), "us_per_step_min": round(1e6 * mn / K, 2),
                      "solves_per_s": round(N * K / med)}), flush=True)

for S in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda") for _ in range(S)]
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    try:
        with torch.cuda.graph(g, stream=cap):
            if S == 1:
                for i in range(K):
                    lib.forward_dynamics_gradient_device(d_in, N, outs[0], stream=cap.cuda_stream)
            else:
                ev0 = torch.cuda.Event()
                ev0.record(cap)
                for s_ in streams:
                    s_.wait_event(ev0)
                for i in range(K):
                    lib.forward_dynamics_gradient_device(d_in, N, outs[i % S], stream=streams[i % S].cuda_stream)
                for s_ in streams:
                    e = torch.cuda.Event()
                    e.record(s_)
                    cap.wait_event(e)
        med, mn = timed(lambda: g.replay())
        print(json.dumps({"mode": "hipGraph, %d chain(s) of kernel nodes" % S, "K": K, "us_per_step_median": round(1e6 * med / K, 2), "us_per_step_min": round(1e6 * mn / K, 2),
                          "solves_per_s": round(N * K / med)}), flush=True)
    except Exception as e:
        print(json.dumps({"mode": "hipGraph %d" % S, "error": str(e)[:200]}), flush=True)
lib.close()
