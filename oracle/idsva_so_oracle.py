"""Second-order derivatives of inverse dynamics (IDSVA-SO): NumPy restatement of the reference's algorithm.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/algorithms/_idsva_so.py:123-915 (gen_idsva_so_inner) statement by statement: the emitted CUDA there IS the
reference's statement of the algorithm (it ships no NumPy oracle for it).  Phases are executed in the reference's order; inside a
phase every thread index i of the reference's parallel loop is visited, so "=" / "+=" and the index-range if/else chains behave as in
the emitted code.  Helper conventions (reference helpers/_spatial_algebra_helpers.py): crm :60-100, crf = -crm^T, icrf :395-447
(icrf(f) x = crf(x) f), outerProduct dest = a b^T (helpers/_lin_alg_helpers.py:103-120).

PARITY UNPINNED: the reference holds no golden vectors, fixtures or tests for this algorithm.  tests/test_idsva_so_oracle.py checks
this restatement against central differences of the pinned first-order oracle (oracle/rbd_oracle.c: rbd_rnea_grad, rbd_minv).

Outputs (n x n x n each, index [i][j][k] = i*n*n + j*n + k as in the reference, :204-208):
    d2tau_dq2[i][j][k]  = d^2 tau_i / dq_j dq_k        d2tau_dqd2[i][j][k] = d^2 tau_i / dqd_j dqd_k
    d2tau_dvdq[i][j][k] = d^2 tau_i / dq_j dqd_k       dM_dq[i][j][k]      = d M_ik / dq_j
"""
import numpy as np


def _sk(x):
    return np.array([[0.0, -x[2], x[1]], [x[2], 0.0, -x[0]], [-x[1], x[0], 0.0]])


def crm(v):
    o = np.zeros((6, 6))
    o[:3, :3] = _sk(v[:3]); o[3:, :3] = _sk(v[3:]); o[3:, 3:] = _sk(v[:3])
    return o


def crf(v):
    return -crm(v).T


def icrf(f):
    """icrf(f) x = crf(x) f.  (The reference lists its 36 entries in the same order as crm's, reference :407-447; read in that order the
    matrix would be the negative of this one, and the finite-difference check of tests/test_idsva_so_oracle.py rejects that reading.)"""
    o = np.zeros((6, 6))
    o[:3, :3] = -_sk(f[:3]); o[:3, 3:] = -_sk(f[3:]); o[3:, :3] = -_sk(f[3:])
    return o


def idsva_so(model, q, qd, qdd, gravity=9.81):
    """model: gridcodegenerator_amd.robot.DuckRobot (numeric tables).  Returns (d2tau_dq2, d2tau_dqd2, d2tau_dvdq, dM_dq)."""
    m = model
    n = m.n
    par = m.parent
    # :216-262 Xup = parent-to-child transforms chained from the base; :265-281 IC = Xup^T I Xup; :284-302 Xdown; :305-311 S = Xdown[:, S_ind]
    Xup, IC, S = [None] * n, [None] * n, np.zeros((n, 6))
    for j in range(n):
        Xj = m.X(j, float(q[j]))
        Xup[j] = Xj if par[j] == -1 else Xj @ Xup[par[j]]
        IC[j] = Xup[j].T @ m.I[j] @ Xup[j]
        S[j] = np.linalg.inv(Xup[j])[:, m.S_index[j]]
    # :314-372 vJ, aJ, v, a, Sd, psid ; :375-396 psidd, IC_v
    v, a, Sd, psid, psidd = (np.zeros((n, 6)) for _ in range(5))
    a_world = np.zeros(6); a_world[5] = gravity
    for j in range(n):
        p = par[j]
        vJ = S[j] * qd[j]
        aJ = S[j] * qdd[j]
        if p != -1:
            aJ = aJ + crm(v[p]) @ vJ
        v[j] = vJ if p == -1 else v[p] + vJ
        Sd[j] = crm(v[j]) @ S[j]
        psid[j] = np.zeros(6) if p == -1 else crm(v[p]) @ S[j]
        a[j] = aJ + a_world if p == -1 else a[p] + aJ
        psidd[j] = crm(a_world) @ S[j] if p == -1 else crm(a[p]) @ S[j] + crm(v[p]) @ psid[j]
    # :399-432 BC = crf(v) IC + icrf(IC v) - IC crm(v), f = IC a + crf(v) IC v   (per body, before the backward accumulation)
    BC, f = [None] * n, np.zeros((n, 6))
    for j in range(n):
        ICv = IC[j] @ v[j]
        BC[j] = crf(v[j]) @ IC[j] + icrf(ICv) - IC[j] @ crm(v[j])
        f[j] = IC[j] @ a[j] + crf(v[j]) @ ICv
    # :438-470 backward accumulation
    for j in range(n - 1, -1, -1):
        if par[j] != -1:
            IC[par[j]] = IC[par[j]] + IC[j]; BC[par[j]] = BC[par[j]] + BC[j]; f[par[j]] = f[par[j]] + f[j]
    # :473-548 per-joint tensors with the composites
    D1, D2, D3, D4, T1, T2, T3, T4, crf_S_IC, IC_S = ([None] * n for _ in range(10))
    psid_Sd = psid + Sd
    for j in range(n):
        IC_S[j] = IC[j] @ S[j]
        D3[j] = crf(S[j]) @ IC[j] + icrf(IC_S[j]) - IC[j] @ crm(S[j])                                   # B_IC_S :492-505
        D2[j] = crf(psid[j]) @ IC[j] + icrf(IC[j] @ psid[j]) - IC[j] @ crm(psid[j])
        D2[j] = D2[j] + crf(S[j]) @ BC[j] - BC[j] @ crm(S[j])                                            # :529-541
        D1[j] = crf(S[j]) @ IC[j] - IC[j] @ crm(S[j])
        D4[j] = icrf(IC_S[j])
        crf_S_IC[j] = crf(S[j]) @ IC[j]
        T1[j] = IC_S[j]
        T2[j] = -BC[j].T @ S[j]
        T3[j] = BC[j] @ psid[j] + IC[j] @ psidd[j] + icrf(f[j]) @ S[j]
        T4[j] = BC[j] @ S[j] + IC[j] @ psid_Sd[j]
    # index lists of the reference's robot object: (joint, ancestor incl. itself) pairs and (joint, ancestor, subtree member incl. itself) triples
    triples = [(j, an, c) for j in range(n) for an in sorted(m.ancestors[j] + [j]) for c in m.subtree[j]]
    pairs = [(j, an) for j in range(n) for an in sorted(m.ancestors[j] + [j])]
    NT = len(triples)
    dq2, dqd2, dvdq, dMdq = (np.zeros((n, n, n)) for _ in range(4))
    bil = lambda x, D, y: float(x @ D @ y)

    def phase(count, body):
        for i in range(count * NT):
            j, an, c = triples[i % NT]
            body(i, j, an, c)

    def t1(i, j, an, c):  # :566-585  t1 = outer(S[j], psid[an])
        x, y = S[j], psid[an]
        if i < NT: dvdq[c, an, j] = -bil(x, D3[c], y)
        elif i < 2 * NT and j != c: dq2[j, c, an] = bil(x, D2[c], y)
        elif i < 3 * NT and j != c: dq2[j, an, c] = bil(x, D2[c], y)
        elif j != c: dvdq[j, an, c] = bil(x, D3[c], y)
    phase(4, t1)

    def t2(i, j, an, c):  # :600-622  t2 = outer(S[j], S[an])
        x, y = S[j], S[an]
        if i < NT and an < j: dqd2[c, j, an] = -bil(x, D3[c], y)
        elif i < NT and j == an: dqd2[c, an, j] = -bil(x, D1[c], y)
        elif i < 2 * NT and j != c: dqd2[j, c, an] = bil(x, D3[c], y)
        elif i < 3 * NT and an < j: dqd2[c, an, j] = -bil(x, D3[c], y)
        elif i < 4 * NT and j != c: dqd2[j, an, c] = bil(x, D3[c], y)
        elif i >= 4 * NT and j != c: dvdq[j, c, an] = bil(x, D2[c], y)
    phase(5, t2)

    def t3(i, j, an, c):  # :637-650  t3 = outer(psid[j], psid[an])
        x, y = psid[j], psid[an]
        if i < NT: dq2[c, an, j] = -bil(x, D3[c], y)
        elif an < j: dq2[c, j, an] = -bil(x, D3[c], y)
    phase(2, t3)

    def t4(i, j, an, c):  # :665-678  t4 = outer(S[j], psidd[an])
        x, y = S[j], psidd[an]
        if i < NT and j != c: dq2[j, c, an] += bil(x, D1[c], y)
        elif j != c: dq2[j, an, c] += bil(x, D1[c], y)
    phase(2, t4)

    def t5(i, j, an, c):  # :693-710  t5 = outer(S[j], (Sd + psid)[an])
        if c != j: dvdq[j, c, an] += bil(S[j], D1[c], psid_Sd[an])
    phase(1, t5)

    def t6(i, j, an, c):  # :725-743  t6 = outer(S[an], psid[j])
        x, y = S[an], psid[j]
        if an < j:
            if i < NT: dvdq[c, j, an] = -bil(x, D3[c], y)
            elif i < 2 * NT: dq2[an, j, c] = bil(x, D2[c], y)
            else: dvdq[an, j, c] = bil(x, D3[c], y)
    phase(3, t6)

    def t7(i, j, an, c):  # :758-772  t7 = outer(S[an], psidd[j])
        if an < j: dq2[an, j, c] += bil(S[an], D1[c], psidd[j])
    phase(1, t7)

    def t8(i, j, an, c):  # :790-818  t8 = outer(S[an], S[j])
        x, y = S[an], S[j]
        if an < j:
            if i < NT: dMdq[an, j, c] = bil(x, D4[c], y)
            elif i < 2 * NT: dMdq[c, j, an] = bil(x, D4[c], y)
            if c != j:
                if i < 3 * NT: dqd2[an, j, c] = bil(x, D3[c], y)
                elif i < 4 * NT: dqd2[an, c, j] = bil(x, D3[c], y)
                elif i < 5 * NT: dvdq[an, c, j] = bil(x, D2[c], y)
        if j != c and i < 6 * NT: dMdq[an, c, j] = bil(x, D1[c], y)
        elif j != c: dMdq[j, c, an] = bil(x, D1[c], y)
    phase(7, t8)

    def t9(i, j, an, c):  # :833-850  t9 = outer(S[an], (Sd + psid)[j])
        if i < NT and an < j and c != j: dvdq[an, c, j] += bil(S[an], D1[c], psid_Sd[j])
        elif an < j and c != j: dq2[an, c, j] = dq2[an, j, c]
    phase(2, t9)

    # :853-873 p1..p6 per (joint, ancestor) pair
    P = {}
    for (j, an) in pairs:
        P[(j, an)] = (crm(psid[an]) @ S[j], crm(psidd[an]) @ S[j], crm(S[an]) @ S[j],
                      crm(psid_Sd[an]) @ S[j] - 2.0 * crm(psid[j]) @ S[an], crm(S[j]) @ S[an],
                      IC_S[j] @ crm(S[an]) + S[an] @ crf_S_IC[j])

    def pfin(i, j, an, c):  # :876-898
        p1, p2, p3, p4, p5, _ = P[(j, an)]
        if i < NT: dq2[c, an, j] += -p1 @ T2[c] + p2 @ T1[c]
        elif an < j:
            if i < 2 * NT: dq2[c, j, an] += -p1 @ T2[c] + p2 @ T1[c]
            elif i < 3 * NT: dvdq[c, j, an] += -p3 @ T2[c] + p4 @ T1[c]
            elif i < 4 * NT: dq2[an, j, c] -= p5 @ T3[c]
            elif i < 5 * NT and c != j: dq2[an, c, j] -= p5 @ T3[c]
            elif i >= 5 * NT: dvdq[an, j, c] -= p5 @ T4[c]
    phase(6, pfin)

    for (j, an) in pairs:  # :901-911
        if an < j: dqd2[an, j, j] = P[(j, an)][5] @ S[j]
    return dq2, dqd2, dvdq, dMdq
