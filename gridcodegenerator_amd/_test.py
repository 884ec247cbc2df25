"""NumPy reference implementations bound to the generator object, like the reference's `_test.py` (README "Additional
Features": test_rnea :109, test_minv :213, test_rnea_grad :490, test_fd_grad :496).  They are DEBUG HELPERS for users who want to
compare device output by hand; nothing on the GPU path (kernels, C ABI, bench, tests) calls them - the tests use the separate
checker in oracle/ and the reference-generated goldens.  Same signatures and conventions as the reference: GRAVITY = -9.81
(negated internally), outputs (c, v, a, f) with 6 x n arrays, dc_du = hstack(dc_dq, dc_dqd), df_du = -Minv dc_du.

Written as plain recursions over the kinematic tree (children fold into parents) on the generator's numeric model.
"""
import numpy as np


def _mxS(s, vec, alpha=1.0):
    """alpha * crm(vec) * e_s"""
    out = np.zeros(6)
    crm = _crm(vec)
    out[:] = crm[:, s] * alpha
    return out


def _crm(v):
    w, l = v[:3], v[3:]
    sk = lambda x: np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]])
    out = np.zeros((6, 6))
    out[:3, :3] = sk(w)
    out[3:, :3] = sk(l)
    out[3:, 3:] = sk(w)
    return out


def _crf(v):
    return -_crm(v).T


def _Xs(self, q):
    m = self.model
    return [m.X(j, float(q[j])) for j in range(m.n)]


def test_rnea(self, q, qd, qdd=None, GRAVITY=-9.81):
    m = self.model
    n = m.n
    X = _Xs(self, q)
    v, a, f = np.zeros((6, n)), np.zeros((6, n)), np.zeros((6, n))
    a_base = np.zeros(6)
    a_base[5] = -GRAVITY
    for j in range(n):  # ids are parent-first
        p, s = m.parent[j], m.S_index[j]
        vp = v[:, p] if p != -1 else np.zeros(6)
        ap = a[:, p] if p != -1 else a_base
        v[:, j] = X[j] @ vp
        v[s, j] += qd[j]
        a[:, j] = X[j] @ ap + _mxS(s, v[:, j], qd[j])
        if qdd is not None:
            a[s, j] += qdd[j]
        f[:, j] = m.I[j] @ a[:, j] + _crf(v[:, j]) @ (m.I[j] @ v[:, j])
    c = np.zeros(n)
    for j in range(n - 1, -1, -1):
        c[j] = f[m.S_index[j], j] + m.damping[j] * qd[j]
        if m.parent[j] != -1:
            f[:, m.parent[j]] += X[j].T @ f[:, j]
    return (c, v, a, f)


def test_minv(self, q, output_dense=True):
    m = self.model
    n = m.n
    X = _Xs(self, q)
    IA = [m.I[j].copy() for j in range(n)]
    F = [np.zeros((6, n)) for _ in range(n)]
    U, Dinv, Minv = np.zeros((n, 6)), np.zeros(n), np.zeros((n, n))
    for j in range(n - 1, -1, -1):
        s, p = m.S_index[j], m.parent[j]
        U[j] = IA[j][:, s]
        Dinv[j] = 1.0 / U[j, s]
        sub = m.subtree[j]
        Minv[j, j] = Dinv[j]
        Minv[j, sub] -= Dinv[j] * F[j][s, sub]
        if p != -1:
            F[j][:, sub] += np.outer(U[j], Minv[j, sub])
            F[p][:, sub] += X[j].T @ F[j][:, sub]
            IA[p] += X[j].T @ (IA[j] - np.outer(U[j], U[j]) * Dinv[j]) @ X[j]
    for j in range(n):
        s, p = m.S_index[j], m.parent[j]
        if p != -1:
            Minv[j, j:] -= Dinv[j] * (U[j] @ X[j]) @ F[p][:, j:]
        F[j][:, j:] = 0.0
        F[j][s, j:] = Minv[j, j:]
        if p != -1:
            F[j][:, j:] += X[j] @ F[p][:, j:]
    if output_dense:
        Minv = np.triu(Minv) + np.triu(Minv, 1).T
    return Minv


def test_rnea_grad(self, q, qd, qdd=None, GRAVITY=-9.81):
    """dc_du (n x 2n) by the same forward-accumulation identity the generated kernels use:
    dc[a][col] = sum_k J_{k,a} . df_k[col] minus the S^T-projection of the propagated mxS(S, f_subtree) corrections."""
    m = self.model
    n = m.n
    X = _Xs(self, q)
    c, v, a, f = test_rnea(self, q, qd, qdd, GRAVITY)  # f is the accumulated subtree force
    a_base = np.zeros(6)
    a_base[5] = -GRAVITY
    dv, da = np.zeros((n, 6, 2 * n)), np.zeros((n, 6, 2 * n))
    dc = np.zeros((n, 2 * n))
    for k in range(n):
        s, p = m.S_index[k], m.parent[k]
        ap = a[:, p] if p != -1 else a_base
        if p != -1:
            dv[k] = X[k] @ dv[p]
            da[k] = X[k] @ da[p]
        dv[k][:, k] += _mxS(s, v[:, k])
        dv[k][s, n + k] += 1.0
        da[k][:, k] += _mxS(s, X[k] @ ap)
        da[k][:, n + k] += _mxS(s, v[:, k])
        for col in range(2 * n):
            da[k][:, col] += _mxS(s, dv[k][:, col], qd[k])
        Iv = m.I[k] @ v[:, k]
        df = m.I[k] @ da[k] + _crf(v[:, k]) @ (m.I[k] @ dv[k])
        for col in range(2 * n):
            df[:, col] += _crf(dv[k][:, col]) @ Iv
        for anc in m.ancestors[k] + [k]:
            dc[anc] += dv[k][:, n + anc] @ df  # J_{k,anc} = d v_k / d qd_anc
    for i in range(n):  # corrections of column q_i travel from joint i to the root
        w = _mxS(m.S_index[i], f[:, i])
        j = i
        while m.parent[j] != -1:
            w = X[j].T @ w
            j = m.parent[j]
            dc[j, i] -= w[m.S_index[j]]
    for i in range(n):
        dc[i, n + i] += m.damping[i]
    return dc


def test_fd_grad(self, q, qd, u, GRAVITY=-9.81):
    c = test_rnea(self, q, qd, None, GRAVITY)[0]
    Minv = test_minv(self, q, True)
    qdd = Minv @ (np.asarray(u) - c)
    return -Minv @ test_rnea_grad(self, q, qd, qdd, GRAVITY)
