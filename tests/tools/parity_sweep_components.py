#!/usr/bin/env python3
"""Randomized parity sweep of the stand-alone kernels (inverse dynamics, its gradient, M^-1, forward dynamics, ABA) on the GPU against the fp64 C oracle.
usage: python tests/tools/parity_sweep_components.py [robot ...]"""
import json, sys
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
from oracle.rbd_oracle import Oracle

robots = sys.argv[1:] or ["iiwa14", "arm6", "hyq", "chain12", "mixed5", "atlas"]
for name in robots:
    robot = RobotModel.from_fixture(name); n = robot.n
    N = 65536 if n <= 12 else 8192
    lib = load(name, max_timesteps=N); orc = Oracle(robot)
    rng = np.random.default_rng(77)
    x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
    qdd = rng.uniform(-5, 5, (N, n)).astype(np.float32)
    st = torch.cuda.current_stream().cuda_stream
    d_x, d_qdd = torch.from_numpy(x).cuda(), torch.from_numpy(qdd).cuda()
    x64, qdd64 = x.astype(np.float64), qdd.astype(np.float64)
    def err(got, ref):
        e = np.abs(got.astype(np.float64) - ref).max(axis=1) / np.abs(ref).max(axis=1)
        return {"max": float(e.max()), "p999": float(np.quantile(e, 0.999)), "median": float(np.median(e))}
    out = {}
    d = torch.empty((N, n), dtype=torch.float32, device="cuda")
    lib.inverse_dynamics_device(d_x, d_qdd, N, d, stream=st); torch.cuda.synchronize()
    out["inverse_dynamics"] = err(d.cpu().numpy(), orc.rnea_batch(x64, qdd64))
    fd_ref = None
    g = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
    lib.inverse_dynamics_gradient_device(d_x, d_qdd, N, g, stream=st); torch.cuda.synchronize()
    out["inverse_dynamics_gradient"] = err(g.cpu().numpy(), orc.rnea_grad_batch(x64, qdd64))
    M = torch.empty((N, n * n), dtype=torch.float32, device="cuda")
    lib.direct_minv_device(d_x, N, M, stream=st); torch.cuda.synchronize()
    Mref = orc.minv_batch(x64)   # device layout: column-major upper triangle, zeros below the diagonal
    out["direct_minv"] = err(M.cpu().numpy(), Mref)
    U = np.transpose(Mref.reshape(N, n, n), (0, 2, 1))            # [row][col], upper triangle
    Mdense = U + np.transpose(np.triu(U, 1), (0, 2, 1))
    # forward dynamics / ABA against qdd = Minv (u - c) from the fp64 oracle
    c0 = orc.rnea_batch(x64, None)
    qref = np.einsum("nij,nj->ni", Mdense, x64[:, 2 * n:] - c0)
    lib.forward_dynamics_device(d_x, N, d, stream=st); torch.cuda.synchronize()
    out["forward_dynamics"] = err(d.cpu().numpy(), qref)
    lib.aba_device(d_x, N, d, stream=st); torch.cuda.synchronize()
    out["aba"] = err(d.cpu().numpy(), qref)
    print(json.dumps({"robot": name, "states": N, **{k: v for k, v in out.items()}}), flush=True)
    lib.close()
