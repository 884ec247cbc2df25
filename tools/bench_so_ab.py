#!/usr/bin/env python3
"""In-process A/B of the second-order kernels (idsva_so with qdd, fdsva_so) of several builds of one robot (tools/build_variant.py): all libraries are
loaded into ONE process, the clocks are warmed once and the builds are timed alternately (box, process and clock state are the same for all of them).
usage: python tools/bench_so_ab.py <robot> <batch> <build-dir | -> [<build-dir> ...]      ('-' = the shipped library)"""
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
d_in = torch.from_numpy(x).cuda(); d_qdd = torch.from_numpy(rng.uniform(-5, 5, (N, n)).astype(np.float32)).cuda()
out = torch.empty((N, 4 * n ** 3), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
run = {"idsva_so": lambda lib: lib.idsva_so_device(d_in, d_qdd, N, out, stream=st), "fdsva_so": lambda lib: lib.fdsva_so_device(d_in, N, out, stream=st)}
t_end = time.perf_counter() + 0.2
while time.perf_counter() < t_end:
    for lib in libs:
        for f in run.values(): f(lib)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
res = {k: [[] for _ in libs] for k in run}
ref = {}
for rep in range(5):
    for kern, f in run.items():
        for i, lib in enumerate(libs):
            for _ in range(2): f(lib)
            e0.record()
            for _ in range(K): f(lib)
            e1.record(); torch.cuda.synchronize()
            res[kern][i].append(round(1e3 * e0.elapsed_time(e1) / K, 2))
            if rep == 0:
                o = out.clone()
                if i == 0: ref[kern] = o
                else: res.setdefault("maxdiff_" + kern, {})[i] = float((o - ref[kern]).abs().max() / ref[kern].abs().max())
for kern in run:
    for i, d in enumerate(dirs):
        r = res[kern][i]
        print(json.dumps({"robot": name, "batch": N, "kernel": kern, "build": d, "us_per_launch_median": sorted(r)[len(r) // 2], "us_per_launch": r,
                          "max_rel_diff_to_first": res.get("maxdiff_" + kern, {}).get(i, 0.0)}))
