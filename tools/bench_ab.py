#!/usr/bin/env python3
"""In-process A/B of forward_dynamics_gradient launches of several builds of one robot (tools/build_variant.py): all libraries are loaded into ONE
process, the clocks are warmed once and the builds are timed alternately, so that box, process and clock state are the same for all of them.
usage: python tools/bench_ab.py <robot> <batch> <build-dir | -> [<build-dir> ...]      ('-' = the shipped library)"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
name, N, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
n = RobotModel.from_fixture(name).n
libs = [load(name, max_timesteps=N, build_dir=None if d == "-" else d) for d in dirs]
rng = np.random.default_rng(0)
x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
d_in = torch.from_numpy(x).cuda()
outs = [torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda") for _ in libs]
st = torch.cuda.current_stream().cuda_stream
t_end = time.perf_counter() + 0.2
while time.perf_counter() < t_end:
    for lib, o in zip(libs, outs):
        for _ in range(10): lib.forward_dynamics_gradient_device(d_in, N, o, stream=st)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 50
res = [[] for _ in libs]
for rep in range(5):
    for i, (lib, o) in enumerate(zip(libs, outs)):
        for _ in range(5): lib.forward_dynamics_gradient_device(d_in, N, o, stream=st)
        e0.record()
        for _ in range(K): lib.forward_dynamics_gradient_device(d_in, N, o, stream=st)
        e1.record(); torch.cuda.synchronize()
        res[i].append(round(1e3 * e0.elapsed_time(e1) / K, 2))
same = [bool(torch.equal(outs[0], o)) for o in outs]
for d, r, s in zip(dirs, res, same):
    print(json.dumps({"robot": name, "batch": N, "build": d, "us_per_launch_median": sorted(r)[len(r) // 2], "us_per_launch": r, "bit_identical_to_first": s}))
