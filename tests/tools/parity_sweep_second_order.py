#!/usr/bin/env python3
"""Parity of idsva_so / fdsva_so on a large batch (LDS re-used by many blocks): GPU vs the NumPy restatements on a random subset of the batch.
usage: python tests/tools/parity_sweep_second_order.py [robot ...]"""
import json, sys
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.robot import DuckRobot
from gridcodegenerator_amd.runtime import load
from oracle.fdsva_so_oracle import fdsva_so
from oracle.idsva_so_oracle import idsva_so
from oracle.rbd_oracle import Oracle

for name in (sys.argv[1:] or ["iiwa14", "arm6", "chain12"]):
    robot = RobotModel.from_fixture(name); n = robot.n; model = DuckRobot(robot); orc = Oracle(robot)
    N = 32768 if n <= 7 else (4096 if n <= 16 else 1024)  # (30 joints: 432 KB per record, the handle's second-order buffers hold 2 485 solves)
    lib = load(name, max_timesteps=N)
    rng = np.random.default_rng(5)
    x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
    qdd = rng.uniform(-5, 5, (N, n)).astype(np.float32)
    st = torch.cuda.current_stream().cuda_stream
    d_x, d_qdd = torch.from_numpy(x).cuda(), torch.from_numpy(qdd).cuda()
    so = torch.empty((N, 4 * n ** 3), dtype=torch.float32, device="cuda"); f2 = torch.empty_like(so)
    lib.idsva_so_device(d_x, d_qdd, N, so, stream=st); lib.fdsva_so_device(d_x, N, f2, stream=st); torch.cuda.synchronize()
    so, f2 = so.cpu().numpy(), f2.cpu().numpy()
    idx = rng.choice(N, 48 if n <= 7 else 12, replace=False)
    e_so, e_f2 = [], []
    for k in idx:
        q, qd, u = (x[k, i * n:(i + 1) * n].astype(np.float64) for i in range(3))
        ref = np.stack(idsva_so(model, q, qd, qdd[k].astype(np.float64))).reshape(4, -1)
        got = so[k].reshape(4, -1).astype(np.float64)
        e_so.append(max(np.abs(got[t] - ref[t]).max() / max(np.abs(ref[t]).max(), 1e-3) for t in range(4)))
        df_du, qdd_fd, Minv, _ = orc.fd_grad(q, qd, u, full=True)
        ref2 = fdsva_so(np.concatenate([t.reshape(-1) for t in idsva_so(model, q, qd, qdd_fd)]), Minv, df_du).reshape(4, -1)
        got2 = f2[k].reshape(4, -1).astype(np.float64)
        e_f2.append(max(np.abs(got2[t] - ref2[t]).max() / max(np.abs(ref2[t]).max(), 1e-3) for t in range(4)))
    print(json.dumps({"robot": name, "batch": N, "checked": len(idx), "finite": bool(np.isfinite(so).all() and np.isfinite(f2).all()),
                      "idsva_so_max_rel_err": float(max(e_so)), "fdsva_so_max_rel_err": float(max(e_f2))}), flush=True)
    lib.close()
