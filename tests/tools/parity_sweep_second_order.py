#!/usr/bin/env python3
"""fp32 error tail of the second-order kernels on the GPU (run from the repository root on the GPU box): idsva_so / fdsva_so on random states of the bench and
the wide distribution, EVERY solve against the NumPy restatements (tests/so_sweep.py; parity unpinned).  One JSON line per robot and distribution.
usage: python tests/tools/parity_sweep_second_order.py [robot[:states] ...]        default: iiwa14:2048 hyq:2048 tree12:2048 atlas:128"""
import json, os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
import so_sweep

if __name__ == "__main__":
    for spec in (sys.argv[1:] or ["iiwa14:2048", "hyq:2048", "tree12:2048", "atlas:128"]):
        name, _, cnt = spec.partition(":")
        n = RobotModel.from_fixture(name).n
        N = int(cnt or (2048 if n <= 12 else 128)) // 2
        lib = load(name, max_timesteps=N)
        for dist, seed in (("bench", 7), ("wide", 8)):
            x, qdd = so_sweep.so_inputs(n, N, dist, seed)
            so, f2 = so_sweep.run_so(torch, lib, x, qdd)
            assert np.isfinite(so).all() and np.isfinite(f2).all()
            e_so, e_f2 = so_sweep.so_errors(name, x, qdd, so, f2)
            print(json.dumps(so_sweep.summarize(name, dist, N, e_so, e_f2)), flush=True)
        lib.close()
