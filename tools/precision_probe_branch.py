#!/usr/bin/env python3
"""Where does the fp32 error tail of the branch-frame forward-dynamics gradient (30-DoF humanoid) come from?  CPU only (test emulation): the T = double
instantiation with ONE stage's results rounded to fp32 (tuning round_probe) on the worst states the GPU sweeps found plus random ones, against the fp64 oracle.
usage: python tools/precision_probe_branch.py <stage>     stage in fp32 | - | chain link vel comp_scan handover comp_I comp_BF t1 t24 M factor t3"""
import sys, json, numpy as np
import os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from emu_harness import emu_library
from gridcodegenerator_amd import RobotModel
from oracle.rbd_oracle import Oracle
stage = sys.argv[1]
name = "atlas"
robot = RobotModel.from_fixture(name); n = robot.n
ws = []
for f in (os.path.join(REPO, "profiles", "r03_parity_sweep.jsonl"),):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            if d["robot"] == name: ws.append(d["worst_state"])
rng = np.random.default_rng(7)
N = 24
xs = np.vstack([np.array(ws, np.float32), np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)])
ref, _ = Oracle(robot).fd_grad_batch(xs.astype(np.float64))
err = lambda out: np.abs(out - ref).max(axis=1) / np.abs(ref).max(axis=1)
if stage == "fp32":
    lib = emu_library(name, max_timesteps=64, tuning={"so_lanes": "off"}); lib.set_launch_dims(0, 64)
    e = err(lib.forward_dynamics_gradient_host(xs))
else:
    st = () if stage == "-" else (stage,)
    lib = emu_library(name, max_timesteps=64, tuning={"so_lanes": "off", "round_probe": st, "allow_wrong_results": True}); lib.set_launch_dims(0, 64)
    e = err(lib.forward_dynamics_gradient_host_f64(xs.astype(np.float64)))
print("%-10s worst-states %s | max %.2e p50 %.2e" % (stage, " ".join("%.1e" % v for v in e[:len(ws)]), e.max(), np.median(e)), flush=True)
