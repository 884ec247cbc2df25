"""Second-order fp32 error sweep (TEST INFRASTRUCTURE): idsva_so / fdsva_so of a robot library on many random states, EVERY solve against the NumPy
restatements oracle/idsva_so_oracle.py / oracle/fdsva_so_oracle.py (parity unpinned - the reference holds no vectors for these algorithms; the restatements
are anchored on finite differences of the pinned first-order oracle, tests/test_idsva_so_oracle.py).  The references are computed by a pool of worker
processes (spawned: they import NumPy and the oracles only, never the GPU runtime).  Used by tests/test_gpu_parity.py and tests/tools/parity_sweep_second_order.py."""
import os

import numpy as np

DISTS = {"bench": (np.pi, 2.0, 10.0, 5.0), "wide": (10 * np.pi, 10.0, 100.0, 50.0)}


def so_inputs(n, N, dist, seed):
    aq, aqd, au, aa = DISTS[dist]
    rng = np.random.default_rng(seed)
    x = np.hstack([rng.uniform(-aq, aq, (N, n)), rng.uniform(-aqd, aqd, (N, n)), rng.uniform(-au, au, (N, n))]).astype(np.float32)
    qdd = rng.uniform(-aa, aa, (N, n)).astype(np.float32)
    return x, qdd


def _tensor_err(got, ref):
    """per solve: max over the four tensors of max|got - ref| / max(max|ref|, 1e-3)"""
    g = got.reshape(4, -1).astype(np.float64)
    r = ref.reshape(4, -1)
    return float(max(np.abs(g[t] - r[t]).max() / max(np.abs(r[t]).max(), 1e-3) for t in range(4)))


def _worker(args):
    name, x, qdd, so_got, f2_got = args
    from gridcodegenerator_amd import RobotModel
    from gridcodegenerator_amd.robot import DuckRobot
    from oracle.fdsva_so_oracle import fdsva_so
    from oracle.idsva_so_oracle import idsva_so
    from oracle.rbd_oracle import Oracle

    robot = RobotModel.from_fixture(name)
    n = robot.n
    model, orc = DuckRobot(robot), Oracle(robot)
    e_so, e_f2 = [], []
    for k in range(x.shape[0]):
        q, qd, u = (x[k, i * n:(i + 1) * n].astype(np.float64) for i in range(3))
        ref = np.concatenate([t.reshape(-1) for t in idsva_so(model, q, qd, qdd[k].astype(np.float64))])
        e_so.append(_tensor_err(so_got[k], ref))
        df_du, qdd_fd, Minv, _ = orc.fd_grad(q, qd, u, full=True)
        ref2 = fdsva_so(np.concatenate([t.reshape(-1) for t in idsva_so(model, q, qd, qdd_fd)]), Minv, df_du)
        e_f2.append(_tensor_err(f2_got[k], ref2))
    return np.array(e_so), np.array(e_f2)


def so_errors(name, x, qdd, so_got, f2_got, nproc=None):
    """Errors of every solve (idsva_so with the given qdd, fdsva_so) - two arrays of len(x)."""
    import multiprocessing as mp

    N = x.shape[0]
    nproc = nproc or max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    nproc = min(nproc, N)
    cuts = np.linspace(0, N, nproc + 1).astype(int)
    jobs = [(name, x[a:b], qdd[a:b], so_got[a:b], f2_got[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    if nproc == 1:
        res = [_worker(j) for j in jobs]
    else:
        with mp.get_context("spawn").Pool(nproc) as pool:
            res = pool.map(_worker, jobs)
    return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])


def run_so(torch, lib, x, qdd):
    """idsva_so (with qdd) and fdsva_so of every state through the C ABI (device-pointer entry points); NaN-prefilled outputs."""
    n, N = lib.n, x.shape[0]
    st = torch.cuda.current_stream().cuda_stream
    d_x, d_qdd = torch.from_numpy(x).cuda(), torch.from_numpy(qdd).cuda()
    so = torch.full((N, 4 * n ** 3), float("nan"), dtype=torch.float32, device="cuda")
    f2 = torch.full((N, 4 * n ** 3), float("nan"), dtype=torch.float32, device="cuda")
    lib.idsva_so_device(d_x, d_qdd, N, so, stream=st)
    lib.fdsva_so_device(d_x, N, f2, stream=st)
    torch.cuda.synchronize()
    return so.cpu().numpy(), f2.cpu().numpy()


def summarize(name, dist, N, e_so, e_f2):
    q = lambda e, p: float(np.quantile(e, p))
    return {"robot": name, "dist": dist, "states": int(N), "every_solve_checked": True,
            "idsva_so": {"max": float(e_so.max()), "p999": q(e_so, 0.999), "p99": q(e_so, 0.99), "median": q(e_so, 0.5), "over_2e-5": int((e_so > 2e-5).sum())},
            "fdsva_so": {"max": float(e_f2.max()), "p999": q(e_f2, 0.999), "p99": q(e_f2, 0.99), "median": q(e_f2, 0.5), "over_2e-5": int((e_f2 > 2e-5).sum())}}
