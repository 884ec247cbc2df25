"""Layout check of the second-order records computed from the library's OWN first-order kernels (no oracle involved).  Test infrastructure,
shared by the emulation (CPU) and the GPU test.

    d2tau_dq2[i][j][k]  = d/dq_k  (dc_i/dq_j)      d2tau_dvdq[i][j][k] = d/dqd_k (dc_i/dq_j)      d2tau_dqd2[i][j][k] = d/dqd_k (dc_i/dqd_j)
    dM_dq[i][j][k]      = d/dq_j  M_ik  (M = columns of inverse_dynamics(q, 0, e_k) at gravity 0)
    d2a_dqdq, d2a_dvdq, d2a_dvdv: the same with df/du;   d2a_dtdq[i][j][k] = d/dq_j Minv_ik.
(reference algorithms/_idsva_so.py:156-159,583-586; _fdsva_so.py:74-81).  fp32 central differences are crude (5e-2 of the tensor's max);
a permuted or transposed layout is an O(1) mismatch."""
import numpy as np


def check_second_order_layout(lib, dev, B=6, h=2e-3, tol=5e-2):
    n = lib.n
    st = dev.stream
    rng = np.random.default_rng(31)
    x = np.hstack([rng.uniform(-np.pi, np.pi, (B, n)), rng.uniform(-2, 2, (B, n)), rng.uniform(-10, 10, (B, n))]).astype(np.float32)
    qdd = rng.uniform(-5, 5, (B, n)).astype(np.float32)

    def dc_du(xx, aa):
        out = dev.full((xx.shape[0], 2 * n * n), np.nan)
        lib.inverse_dynamics_gradient_device(dev.arr(xx), dev.arr(aa), xx.shape[0], out, stream=st)
        return dev.host(out).astype(np.float64).reshape(-1, 2 * n, n)     # [b][col][row i]

    def df_du(xx):
        out = dev.full((xx.shape[0], 2 * n * n), np.nan)
        lib.forward_dynamics_gradient_device(dev.arr(xx), xx.shape[0], out, stream=st)
        return dev.host(out).astype(np.float64).reshape(-1, 2 * n, n)

    def mass_matrix(xx):   # M[:, k] = ID(q, 0, e_k) with gravity 0
        M = np.zeros((xx.shape[0], n, n))
        x0 = xx.copy()
        x0[:, n:2 * n] = 0
        for k in range(n):
            e = np.zeros((xx.shape[0], n), np.float32)
            e[:, k] = 1
            out = dev.full((xx.shape[0], n), np.nan)
            lib.inverse_dynamics_device(dev.arr(x0), dev.arr(e), xx.shape[0], out, gravity=0.0, stream=st)
            M[:, :, k] = dev.host(out)
        return M

    def minv(xx):
        out = dev.full((xx.shape[0], n * n), np.nan)
        lib.direct_minv_device(dev.arr(xx), xx.shape[0], out, stream=st)
        U = dev.host(out).astype(np.float64).reshape(-1, n, n)              # [b][col][row], upper triangle
        full = np.zeros_like(U)
        for b in range(U.shape[0]):
            up = U[b].T
            full[b] = np.triu(up) + np.triu(up, 1).T
        return full

    so = dev.full((B, 4 * n ** 3), np.nan)
    lib.idsva_so_device(dev.arr(x), dev.arr(qdd), B, so, stream=st)
    df2 = dev.full((B, 4 * n ** 3), np.nan)
    lib.fdsva_so_device(dev.arr(x), B, df2, stream=st)   # (takes the idsva_so tensors at qdd = FD(q, qd, u), which df_du's differences follow by themselves)
    so = dev.host(so).astype(np.float64).reshape(B, 4, n, n, n)
    df2 = dev.host(df2).astype(np.float64).reshape(B, 4, n, n, n)
    assert np.isfinite(so).all() and np.isfinite(df2).all()
    fd_so, fd_df2 = np.zeros_like(so), np.zeros_like(df2)
    for k in range(n):
        for is_v in (0, 1):
            xp, xm = x.copy(), x.copy()
            xp[:, is_v * n + k] += h
            xm[:, is_v * n + k] -= h
            d_id = (dc_du(xp, qdd) - dc_du(xm, qdd)) / (2 * h)                   # [b][col][i]
            d_fd = (df_du(xp) - df_du(xm)) / (2 * h)
            if not is_v:
                fd_so[:, 0, :, :, k] = d_id[:, :n, :].transpose(0, 2, 1)          # d/dq_k (dc_i/dq_j)
                fd_df2[:, 0, :, :, k] = d_fd[:, :n, :].transpose(0, 2, 1)
                fd_so[:, 3, :, k, :] = (mass_matrix(xp) - mass_matrix(xm)) / (2 * h)   # dM_dq[i][j = k][k'] = d M_ik' / dq_k
                fd_df2[:, 3, :, k, :] = (minv(xp) - minv(xm)) / (2 * h)
            else:
                fd_so[:, 2, :, :, k] = d_id[:, :n, :].transpose(0, 2, 1)          # d/dqd_k (dc_i/dq_j)
                fd_so[:, 1, :, :, k] = d_id[:, n:, :].transpose(0, 2, 1)          # d/dqd_k (dc_i/dqd_j)
                fd_df2[:, 2, :, :, k] = d_fd[:, :n, :].transpose(0, 2, 1)
                fd_df2[:, 1, :, :, k] = d_fd[:, n:, :].transpose(0, 2, 1)
    names = (("d2tau_dq2", "d2tau_dqd2", "d2tau_dvdq", "dM_dq"), ("d2a_dqdq", "d2a_dvdv", "d2a_dvdq", "d2a_dtdq"))
    report = {}
    for which, (got, fd) in enumerate(((so, fd_so), (df2, fd_df2))):
        for t in range(4):
            scale = np.abs(got[:, t]).max(axis=(1, 2, 3), keepdims=True)
            err = (np.abs(got[:, t] - fd[:, t]) / scale).max()
            report[names[which][t]] = err
            assert err < tol, (names[which][t], err)
            if t in (2, 3):  # the check must be able to tell a transposed layout: these tensors are not symmetric in (j, k)
                assert (np.abs(got[:, t] - got[:, t].transpose(0, 1, 3, 2)) / scale).max() > 0.2, names[which][t]
    return report
