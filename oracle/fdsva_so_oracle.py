"""Second-order derivatives of forward dynamics (FDSVA-SO): NumPy restatement of the reference's contraction.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/algorithms/_fdsva_so.py:3-85 (gen_fdsva_so_inner) statement by statement, with the reference's flat indexing;
the inputs are what the reference's fdsva_so_device (:121-157) feeds it: the idsva_so tensors evaluated at qdd = FD(q, qd, u), the
dense symmetric M^-1 and df/du = [df/dq | df/dqd] (column-major n x n blocks).

PARITY UNPINNED (no reference vectors exist for it): tests/test_idsva_so_oracle.py anchors it on central differences of the pinned
first-order forward-dynamics-gradient oracle (oracle/rbd_oracle.c: rbd_fd_grad).

Output df2 = [d2a_dqdq | d2a_dvdv | d2a_dvdq | d2a_dtdq], each n x n x n, flat index i*n*n + j*n + k:
    d2a_dqdq[i][j][k] = d^2 qdd_i / dq_j dq_k      d2a_dvdv[i][j][k] = d^2 qdd_i / dqd_j dqd_k
    d2a_dvdq[i][j][k] = d^2 qdd_i / dq_j dqd_k     d2a_dtdq[i][j][k] = d^2 qdd_i / dq_j dtau_k  (= d Minv_ik / dq_j)
"""
import numpy as np


def fdsva_so(idsva_so_flat, Minv, df_du):
    """idsva_so_flat: 4 n^3 values [d2tau_dq2 | d2tau_dqd2 | d2tau_dvdq | dM_dq]; Minv: dense n x n; df_du: n x 2n.  Returns the 4 n^3 values of df2."""
    n = Minv.shape[0]
    n2, n3 = n * n, n * n * n
    so = np.asarray(idsva_so_flat, dtype=np.float64).reshape(-1)
    d2tau_dqdq, d2tau_dvdv, d2tau_dvdq, dM_dq = so[:n3], so[n3:2 * n3], so[2 * n3:3 * n3], so[3 * n3:]
    s_Minv = np.asarray(Minv, dtype=np.float64).T.reshape(-1).copy()        # column-major n x n (symmetric anyway)
    s_df_dq = np.asarray(df_du[:, :n], dtype=np.float64).T.reshape(-1)       # column-major: [n*j + L] = d qdd_L / d q_j
    s_df_dqd = np.asarray(df_du[:, n:], dtype=np.float64).T.reshape(-1)
    inner_dq, inner_cross, inner_tau, rot_dq = (np.zeros(n3) for _ in range(4))
    for ind in range(n3):                                                     # :52-61
        i, j, k = ind // n2 % n, ind // n % n, ind % n
        inner_dq[ind] = dM_dq[n2 * i + n * k: n2 * i + n * k + n] @ s_df_dq[n * j: n * j + n]
        rot_dq[i * n2 + k * n + j] = inner_dq[ind]
    for ind in range(3 * n3):                                                 # :64-71
        i, j, k = ind // n2 % n, ind // n % n, ind % n
        if ind < n3:
            inner_dq[ind] += rot_dq[ind] + d2tau_dqdq[ind]
        elif ind < 2 * n3:
            inner_cross[i * n2 + k * n + j] = dM_dq[n2 * i + n * k: n2 * i + n * k + n] @ s_df_dqd[n * j: n * j + n] + d2tau_dvdq[i * n2 + k * n + j]
        else:
            inner_tau[i * n2 + k * n + j] = dM_dq[n2 * i + n * k: n2 * i + n * k + n] @ s_Minv[n * j: n * j + n]
    out = np.zeros(4 * n3)
    d2a_dqdq, d2a_dvdv, d2a_dvdq, d2a_dtdq = out[:n3], out[n3:2 * n3], out[2 * n3:3 * n3], out[3 * n3:]
    col = lambda i: s_Minv[i::n][:n]                                          # &s_Minv[i] with stride n
    for ind in range(n3):                                                     # :74-81
        i, j, k = ind // n2 % n, ind // n % n, ind % n
        sl = slice(j + k * n, None, n2)
        d2a_dqdq[i * n2 + j + k * n] = -(col(i) @ inner_dq[sl][:n])
        d2a_dvdq[i * n2 + j + k * n] = -(col(i) @ inner_cross[sl][:n])
        d2a_dvdv[i * n2 + j + k * n] = -(col(i) @ d2tau_dvdv[sl][:n])
        d2a_dtdq[i * n2 + j + k * n] = -(col(i) @ inner_tau[sl][:n])
    return out
