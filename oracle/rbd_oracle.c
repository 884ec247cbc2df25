/*
 * rbd_oracle.c - CPU restatement of the reference's NumPy oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the HIP path; it is never part of the product path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load it.
 *
 * Each function follows the reference file /root/reference/_test.py (cited per function).  The oracle
 * is PINNED: tests/test_oracle.py checks it against golden vectors produced by importing and running
 * the reference's own functions in the build container (tests/golden/make_goldens.py).
 *
 * Precision: compiled twice, -DRBD_REAL=double (symbols *_f64) and -DRBD_REAL=float (symbols *_f32).
 * Conventions: 6-vectors angular-first; X_i(q) = X_J(q_i) * X_tree_i ; matrices row-major [row*6+col];
 * gravity is passed POSITIVE (9.81): a_base = X[:,5]*gravity, i.e. the reference's gravity_vec[5] = -GRAVITY
 * with GRAVITY=-9.81 (_test.py:13-14).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef RBD_REAL
#define RBD_REAL double
#define RBD_SUFFIX _f64
#endif
#define RBD_CAT2(a, b) a##b
#define RBD_CAT(a, b) RBD_CAT2(a, b)
#define FN(name) RBD_CAT(name, RBD_SUFFIX)
typedef RBD_REAL real;

#define RBD_MAX_N 64

typedef struct {
    int n;
    const int *parent;   /* [n], -1 = base */
    const int *S_index;  /* [n], 0..2 revolute x/y/z, 3..5 prismatic x/y/z */
    const real *X_tree;  /* [n][36] row-major */
    const real *I;       /* [n][36] row-major */
    const real *damping; /* [n] */
} FN(rbd_model);
typedef FN(rbd_model) model_t;

/* ---- joint transform X(q) = X_J(q) X_tree (robot front end; matches get_Xmat_Func_by_id) ---- */
static void Xmat(const model_t *m, int i, real q, real *X) {
    const real *T = m->X_tree + 36 * i;
    int s = m->S_index[i];
    real XJ[36];
    memset(XJ, 0, sizeof(XJ));
    if (s < 3) {
        real c = (real)cos((double)q), sn = (real)sin((double)q);
        real E[9];
        if (s == 0) { real e[9] = {1, 0, 0, 0, c, sn, 0, -sn, c}; memcpy(E, e, sizeof(E)); }
        else if (s == 1) { real e[9] = {c, 0, -sn, 0, 1, 0, sn, 0, c}; memcpy(E, e, sizeof(E)); }
        else { real e[9] = {c, sn, 0, -sn, c, 0, 0, 0, 1}; memcpy(E, e, sizeof(E)); }
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) { XJ[r * 6 + cc] = E[r * 3 + cc]; XJ[(r + 3) * 6 + cc + 3] = E[r * 3 + cc]; }
    } else {
        real d[3] = {0, 0, 0};
        d[s - 3] = q;
        for (int r = 0; r < 6; r++) XJ[r * 6 + r] = 1;
        /* bottom-left = -skew(d) */
        XJ[3 * 6 + 1] = d[2];  XJ[3 * 6 + 2] = -d[1];
        XJ[4 * 6 + 0] = -d[2]; XJ[4 * 6 + 2] = d[0];
        XJ[5 * 6 + 0] = d[1];  XJ[5 * 6 + 1] = -d[0];
    }
    for (int r = 0; r < 6; r++) for (int c2 = 0; c2 < 6; c2++) {
        real acc = 0;
        for (int k = 0; k < 6; k++) acc += XJ[r * 6 + k] * T[k * 6 + c2];
        X[r * 6 + c2] = acc;
    }
}

static void matvec6(const real *M, const real *x, real *y) { /* y = M x */
    for (int r = 0; r < 6; r++) { real acc = 0; for (int k = 0; k < 6; k++) acc += M[r * 6 + k] * x[k]; y[r] = acc; }
}
static void matTvec6(const real *M, const real *x, real *y) { /* y = M^T x */
    for (int r = 0; r < 6; r++) { real acc = 0; for (int k = 0; k < 6; k++) acc += M[k * 6 + r] * x[k]; y[r] = acc; }
}

/* mxS: motion cross product column crm(vec)*S*alpha  (_test.py:522-608, mx0..mx5) */
static void mxS(int s, const real *vec, real alpha, real *out) {
    for (int r = 0; r < 6; r++) out[r] = 0;
    switch (s) {
    case 0: out[1] = vec[2] * alpha; out[2] = -vec[1] * alpha; out[4] = vec[5] * alpha; out[5] = -vec[4] * alpha; break;
    case 1: out[0] = -vec[2] * alpha; out[2] = vec[0] * alpha; out[3] = -vec[5] * alpha; out[5] = vec[3] * alpha; break;
    case 2: out[0] = vec[1] * alpha; out[1] = -vec[0] * alpha; out[3] = vec[4] * alpha; out[4] = -vec[3] * alpha; break;
    case 3: out[4] = vec[2] * alpha; out[5] = -vec[1] * alpha; break;
    case 4: out[3] = -vec[2] * alpha; out[5] = vec[0] * alpha; break;
    case 5: out[3] = vec[1] * alpha; out[4] = -vec[0] * alpha; break;
    }
}

/* fxv: force cross product Fx(a)*b  (_test.py:649-664) */
static void fxv(const real *a, const real *b, real *r) {
    r[0] = -a[2] * b[1] + a[1] * b[2] - a[5] * b[4] + a[4] * b[5];
    r[1] = a[2] * b[0] - a[0] * b[2] + a[5] * b[3] - a[3] * b[5];
    r[2] = -a[1] * b[0] + a[0] * b[1] - a[4] * b[3] + a[3] * b[4];
    r[3] = -a[2] * b[4] + a[1] * b[5];
    r[4] = a[2] * b[3] - a[0] * b[5];
    r[5] = -a[1] * b[3] + a[0] * b[4];
}

static int is_ancestor(const model_t *m, int anc, int j) { /* anc in ancestors(j) */
    int p = m->parent[j];
    while (p != -1) { if (p == anc) return 1; p = m->parent[p]; }
    return 0;
}

/* ---------------- RNEA: _test.py:5-115 (test_rnea_fpass, test_rnea_bpass, test_rnea) ----------------
 * outputs: c[n]; v,a,f as [n][6] (f is the accumulated force after the backward pass, as in the reference
 * where bpass updates f in place). qdd may be NULL. X (n*36) is an optional cache of the joint transforms. */
static void rnea_core(const model_t *m, const real *Xs, const real *qd, const real *qdd, real gravity,
                      real *c, real *v, real *a, real *f) {
    int n = m->n;
    /* forward pass: parents precede children in id order, equivalent to the BFS-level sweep (_test.py:24-62) */
    for (int i = 0; i < n; i++) {
        const real *X = Xs + 36 * i;
        int s = m->S_index[i], p = m->parent[i];
        real *vi = v + 6 * i, *ai = a + 6 * i;
        if (p == -1) {
            for (int r = 0; r < 6; r++) { vi[r] = 0; ai[r] = X[r * 6 + 5] * gravity; } /* X * [0..0,g] */
            vi[s] += qd[i];
            if (qdd) ai[s] += qdd[i];
        } else {
            matvec6(X, v + 6 * p, vi);
            matvec6(X, a + 6 * p, ai);
            vi[s] += qd[i];
            if (qdd) ai[s] += qdd[i];
            real mv[6];
            mxS(s, vi, qd[i], mv);
            for (int r = 0; r < 6; r++) ai[r] += mv[r];
        }
    }
    for (int i = 0; i < n; i++) { /* f = I a + fx(v) I v  (_test.py:64-67) */
        real Iv[6], Ia[6], t[6];
        matvec6(m->I + 36 * i, v + 6 * i, Iv);
        matvec6(m->I + 36 * i, a + 6 * i, Ia);
        fxv(v + 6 * i, Iv, t);
        for (int r = 0; r < 6; r++) f[6 * i + r] = Ia[r] + t[r];
    }
    /* backward pass (_test.py:88-101) children before parents */
    for (int i = n - 1; i >= 0; i--) {
        int p = m->parent[i];
        c[i] = f[6 * i + m->S_index[i]];
        if (p != -1) {
            real t[6];
            matTvec6(Xs + 36 * i, f + 6 * i, t);
            for (int r = 0; r < 6; r++) f[6 * p + r] += t[r];
        }
    }
    for (int i = 0; i < n; i++) c[i] += m->damping[i] * qd[i]; /* velocity damping (_test.py:103-105) */
}

void FN(rbd_rnea)(const model_t *m, const real *q, const real *qd, const real *qdd, real gravity,
                  real *c, real *v, real *a, real *f) {
    real Xs[36 * RBD_MAX_N];
    for (int i = 0; i < m->n; i++) Xmat(m, i, q[i], Xs + 36 * i);
    rnea_core(m, Xs, qd, qdd, gravity, c, v, a, f);
}

/* ---------------- Minv: _test.py:117-226 (test_minv_bpass, test_minv_fpass, test_densify_Minv) -------
 * Minv is n x n row-major.  dense != 0 fills the lower triangle by symmetry. */
static void minv_core(const model_t *m, const real *Xs, real *Minv, int dense) {
    int n = m->n;
    /* C99 variable-length arrays: no allocator traffic in the timed multi-threaded baseline */
    real F[(size_t)n * 6 * n]; /* F[i][row][col] */
    real U[(size_t)n * 6];
    real IA[(size_t)n * 36];
    real Dinv[RBD_MAX_N];
    memset(F, 0, sizeof(F));
    memset(U, 0, sizeof(U));
    memcpy(IA, m->I, (size_t)n * 36 * sizeof(real));
    memset(Minv, 0, (size_t)n * n * sizeof(real));
#define Fm(i, r, c) F[((size_t)(i) * 6 + (r)) * n + (c)]
    for (int i = n - 1; i >= 0; i--) { /* backward pass (_test.py:132-183) */
        int s = m->S_index[i], p = m->parent[i];
        const real *X = Xs + 36 * i;
        real *IAi = IA + 36 * i;
        for (int r = 0; r < 6; r++) U[6 * i + r] = IAi[r * 6 + s];
        Dinv[i] = (real)1 / U[6 * i + s];
        Minv[i * n + i] = Dinv[i];
        /* subtree(i) = { j >= i : i == j or i in ancestors(j) } */
        for (int j = i; j < n; j++) {
            if (j != i && !is_ancestor(m, i, j)) continue;
            Minv[i * n + j] -= Dinv[i] * Fm(i, s, j);
        }
        if (p != -1) {
            for (int j = i; j < n; j++) {
                if (j != i && !is_ancestor(m, i, j)) continue;
                real col[6], t[6];
                for (int r = 0; r < 6; r++) { Fm(i, r, j) += U[6 * i + r] * Minv[i * n + j]; col[r] = Fm(i, r, j); }
                matTvec6(X, col, t);
                for (int r = 0; r < 6; r++) Fm(p, r, j) += t[r];
            }
            real Ia[36], IaX[36];
            for (int r = 0; r < 6; r++) for (int c2 = 0; c2 < 6; c2++) Ia[r * 6 + c2] = IAi[r * 6 + c2] - U[6 * i + r] * Dinv[i] * U[6 * i + c2];
            for (int r = 0; r < 6; r++) for (int c2 = 0; c2 < 6; c2++) { real acc = 0; for (int k = 0; k < 6; k++) acc += Ia[r * 6 + k] * X[k * 6 + c2]; IaX[r * 6 + c2] = acc; }
            for (int r = 0; r < 6; r++) for (int c2 = 0; c2 < 6; c2++) { real acc = 0; for (int k = 0; k < 6; k++) acc += X[k * 6 + r] * IaX[k * 6 + c2]; IA[36 * p + r * 6 + c2] += acc; }
        }
    }
    for (int i = 0; i < n; i++) { /* forward pass (_test.py:192-200), serial over joints */
        int s = m->S_index[i], p = m->parent[i];
        const real *X = Xs + 36 * i;
        if (p != -1) {
            real UX[6];
            matTvec6(X, U + 6 * i, UX); /* (U^T X)^T */
            for (int j = i; j < n; j++) {
                real acc = 0;
                for (int r = 0; r < 6; r++) acc += UX[r] * Fm(p, r, j);
                Minv[i * n + j] -= Dinv[i] * acc;
            }
        }
        for (int j = i; j < n; j++) {
            real col[6] = {0, 0, 0, 0, 0, 0}, t[6] = {0, 0, 0, 0, 0, 0};
            if (p != -1) { for (int r = 0; r < 6; r++) col[r] = Fm(p, r, j); matvec6(X, col, t); }
            for (int r = 0; r < 6; r++) Fm(i, r, j) = t[r];
            Fm(i, s, j) += Minv[i * n + j];
        }
    }
    if (dense) for (int r = 0; r < n; r++) for (int c2 = 0; c2 < r; c2++) Minv[r * n + c2] = Minv[c2 * n + r];
#undef Fm
}

void FN(rbd_minv)(const model_t *m, const real *q, real *Minv, int dense) {
    real Xs[36 * RBD_MAX_N];
    for (int i = 0; i < m->n; i++) Xmat(m, i, q[i], Xs + 36 * i);
    minv_core(m, Xs, Minv, dense);
}

/* ---------------- RNEA gradient: _test.py:229-494 (test_rnea_grad_inner, test_rnea_grad) -------------
 * dc_du is n x 2n row-major: [dc/dq | dc/dqd]; v,a,f are the RNEA outputs for (q,qd,qdd). */
static void rnea_grad_core(const model_t *m, const real *Xs, const real *qd, const real *v, const real *a, const real *f,
                           real gravity, real *dc_du) {
    int n = m->n;
    size_t sz = (size_t)6 * n * n;
    real dbuf[sz * 6];
    memset(dbuf, 0, sizeof(dbuf));
    real *dv_dq = dbuf;
    real *dv_dqd = dv_dq + sz, *da_dq = dv_dq + 2 * sz, *da_dqd = dv_dq + 3 * sz, *df_dq = dv_dq + 4 * sz, *df_dqd = dv_dq + 5 * sz;
#define D(arr, row, col, ind) arr[((size_t)(ind) * n + (col)) * 6 + (row)] /* column vectors contiguous */
    real MxXv[6 * RBD_MAX_N], MxXa[6 * RBD_MAX_N], Mxv[6 * RBD_MAX_N], Mxf[6 * RBD_MAX_N], Iv[6 * RBD_MAX_N];
    for (int i = 0; i < n; i++) { /* temps (_test.py:284-311) */
        const real *X = Xs + 36 * i;
        int p = m->parent[i], s = m->S_index[i];
        real Xv[6], Xa[6];
        if (p != -1) { matvec6(X, v + 6 * p, Xv); matvec6(X, a + 6 * p, Xa); }
        else { for (int r = 0; r < 6; r++) { Xv[r] = 0; Xa[r] = X[r * 6 + 5] * gravity; } }
        matvec6(m->I + 36 * i, v + 6 * i, Iv + 6 * i);
        mxS(s, Xv, 1, MxXv + 6 * i);
        mxS(s, Xa, 1, MxXa + 6 * i);
        mxS(s, v + 6 * i, 1, Mxv + 6 * i);
        mxS(s, f + 6 * i, 1, Mxf + 6 * i);
    }
    for (int i = 0; i < n; i++) { /* dv/du (_test.py:327-344); id order == BFS-level order for dependencies */
        const real *X = Xs + 36 * i;
        int p = m->parent[i], s = m->S_index[i];
        for (int col = 0; col < i; col++) {
            if (!is_ancestor(m, col, i)) continue;
            matvec6(X, &D(dv_dq, 0, col, p), &D(dv_dq, 0, col, i));
            matvec6(X, &D(dv_dqd, 0, col, p), &D(dv_dqd, 0, col, i));
        }
        if (p != -1) for (int r = 0; r < 6; r++) D(dv_dq, r, i, i) += MxXv[6 * i + r];
        D(dv_dqd, s, i, i) += 1;
    }
    for (int i = 0; i < n; i++) { /* da/du part 1 (_test.py:352-362) */
        int s = m->S_index[i];
        for (int col = 0; col <= i; col++) {
            if (col != i && !is_ancestor(m, col, i)) continue;
            mxS(s, &D(dv_dq, 0, col, i), qd[i], &D(da_dq, 0, col, i));
            mxS(s, &D(dv_dqd, 0, col, i), qd[i], &D(da_dqd, 0, col, i));
            if (col == i) for (int r = 0; r < 6; r++) { D(da_dq, r, col, i) += MxXa[6 * i + r]; D(da_dqd, r, col, i) += Mxv[6 * i + r]; }
        }
    }
    for (int i = 0; i < n; i++) { /* da/du += X da_parent/du (_test.py:370-381) */
        int p = m->parent[i];
        if (p == -1) continue;
        const real *X = Xs + 36 * i;
        for (int col = 0; col <= i; col++) {
            if (col != i && !is_ancestor(m, col, i)) continue;
            real t[6];
            matvec6(X, &D(da_dq, 0, col, p), t);
            for (int r = 0; r < 6; r++) D(da_dq, r, col, i) += t[r];
            matvec6(X, &D(da_dqd, 0, col, p), t);
            for (int r = 0; r < 6; r++) D(da_dqd, r, col, i) += t[r];
        }
    }
    for (int i = 0; i < n; i++) { /* df/du (_test.py:389-424): fx(dv)Iv + I da + (fx(v) I) dv */
        const real *I = m->I + 36 * i;
        real FxvI[36]; /* FxvI[:,col] = fxv(v, I[:,col]) */
        for (int col = 0; col < 6; col++) {
            real Icol[6], t[6];
            for (int r = 0; r < 6; r++) Icol[r] = I[r * 6 + col];
            fxv(v + 6 * i, Icol, t);
            for (int r = 0; r < 6; r++) FxvI[r * 6 + col] = t[r];
        }
        for (int col = 0; col <= i; col++) {
            if (col != i && !is_ancestor(m, col, i)) continue;
            real t1[6], t2[6], t3[6];
            fxv(&D(dv_dq, 0, col, i), Iv + 6 * i, t1); matvec6(I, &D(da_dq, 0, col, i), t2); matvec6(FxvI, &D(dv_dq, 0, col, i), t3);
            for (int r = 0; r < 6; r++) D(df_dq, r, col, i) = t1[r] + t2[r] + t3[r];
            fxv(&D(dv_dqd, 0, col, i), Iv + 6 * i, t1); matvec6(I, &D(da_dqd, 0, col, i), t2); matvec6(FxvI, &D(dv_dqd, 0, col, i), t3);
            for (int r = 0; r < 6; r++) D(df_dqd, r, col, i) = t1[r] + t2[r] + t3[r];
        }
    }
    for (int i = n - 1; i >= 0; i--) { /* backward pass (_test.py:450-470) */
        int p = m->parent[i];
        if (p == -1) continue;
        const real *X = Xs + 36 * i;
        real Xmxf[6];
        matTvec6(X, Mxf + 6 * i, Xmxf);
        for (int col = 0; col < n; col++) {
            int in_sub = (col >= i) && (col == i || is_ancestor(m, i, col));
            if (!in_sub && !is_ancestor(m, col, i)) continue;
            real t[6];
            matTvec6(X, &D(df_dq, 0, col, i), t);
            for (int r = 0; r < 6; r++) D(df_dq, r, col, p) += t[r];
            matTvec6(X, &D(df_dqd, 0, col, i), t);
            for (int r = 0; r < 6; r++) D(df_dqd, r, col, p) += t[r];
            if (col == i) for (int r = 0; r < 6; r++) D(df_dq, r, col, p) -= Xmxf[r];
        }
    }
    memset(dc_du, 0, (size_t)n * 2 * n * sizeof(real));
    for (int i = 0; i < n; i++) { /* extraction (_test.py:479-486) incl. damping on diag(dc/dqd) */
        int s = m->S_index[i];
        for (int col = 0; col < n; col++) {
            int in_sub = (col >= i) && (col == i || is_ancestor(m, i, col));
            if (!in_sub && !is_ancestor(m, col, i)) continue;
            dc_du[i * 2 * n + col] = D(df_dq, s, col, i);
            dc_du[i * 2 * n + n + col] = D(df_dqd, s, col, i) + (i == col ? m->damping[i] : (real)0);
        }
    }
#undef D
}

void FN(rbd_rnea_grad)(const model_t *m, const real *q, const real *qd, const real *qdd, real gravity, real *dc_du) {
    int n = m->n;
    real Xs[36 * RBD_MAX_N], c[RBD_MAX_N], v[6 * RBD_MAX_N], a[6 * RBD_MAX_N], f[6 * RBD_MAX_N];
    for (int i = 0; i < n; i++) Xmat(m, i, q[i], Xs + 36 * i);
    rnea_core(m, Xs, qd, qdd, gravity, c, v, a, f);
    rnea_grad_core(m, Xs, qd, v, a, f, gravity, dc_du);
}

/* ---------------- FD gradient: _test.py:496-520 (test_fd_grad) -----------------------------------
 * df_du = -Minv * dc_du, n x 2n row-major.  Optional outputs (may be NULL): qdd[n], Minv[n*n] dense, dc_du. */
void FN(rbd_fd_grad)(const model_t *m, const real *q, const real *qd, const real *u, real gravity,
                     real *df_du, real *qdd_out, real *Minv_out, real *dc_du_out) {
    int n = m->n;
    real Xs[36 * RBD_MAX_N], c[RBD_MAX_N], v[6 * RBD_MAX_N], a[6 * RBD_MAX_N], f[6 * RBD_MAX_N], qdd[RBD_MAX_N];
    real Minv[(size_t)n * n];
    real dc_du[(size_t)n * 2 * n];
    for (int i = 0; i < n; i++) Xmat(m, i, q[i], Xs + 36 * i);
    rnea_core(m, Xs, qd, NULL, gravity, c, v, a, f);
    minv_core(m, Xs, Minv, 1);
    for (int r = 0; r < n; r++) { real acc = 0; for (int k = 0; k < n; k++) acc += Minv[r * n + k] * (u[k] - c[k]); qdd[r] = acc; }
    rnea_core(m, Xs, qd, qdd, gravity, c, v, a, f);
    rnea_grad_core(m, Xs, qd, v, a, f, gravity, dc_du);
    for (int r = 0; r < n; r++) for (int col = 0; col < 2 * n; col++) {
        real acc = 0;
        for (int k = 0; k < n; k++) acc += Minv[r * n + k] * dc_du[k * 2 * n + col];
        df_du[r * 2 * n + col] = -acc;
    }
    if (qdd_out) memcpy(qdd_out, qdd, (size_t)n * sizeof(real));
    if (Minv_out) memcpy(Minv_out, Minv, (size_t)n * n * sizeof(real));
    if (dc_du_out) memcpy(dc_du_out, dc_du, (size_t)n * 2 * n * sizeof(real));
}

/* ---------------- batch drivers in the DEVICE layouts --------------------
 * in : q_qd_u[k*stride + {0..n | n..2n | 2n..3n}]   out: df_du[k*2n^2 + col*n + row]  (column-major n x 2n)
 * nthreads <= 0 -> all OpenMP threads.  Returns the thread count actually used. */
int FN(rbd_fd_grad_batch)(const model_t *m, int N, const real *q_qd_u, int stride, real gravity, real *df_du_dev, int nthreads) {
    int n = m->n, used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = (nthreads > 0) ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(static)
#endif
    for (int k = 0; k < N; k++) {
        real out[2 * RBD_MAX_N * RBD_MAX_N];
        const real *in = q_qd_u + (size_t)k * stride;
        FN(rbd_fd_grad)(m, in, in + n, in + 2 * n, gravity, out, NULL, NULL, NULL);
        real *dst = df_du_dev + (size_t)k * 2 * n * n;
        for (int col = 0; col < 2 * n; col++) for (int r = 0; r < n; r++) dst[col * n + r] = out[r * 2 * n + col];
    }
    return used;
}

/* c[k*n+i] from q_qd[k*stride + ...] (+ qdd[k*n+i] if qdd != NULL) */
void FN(rbd_rnea_batch)(const model_t *m, int N, const real *q_qd, int stride, const real *qdd, real gravity, real *c_out) {
    int n = m->n;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int k = 0; k < N; k++) {
        real v[6 * RBD_MAX_N], a[6 * RBD_MAX_N], f[6 * RBD_MAX_N];
        const real *in = q_qd + (size_t)k * stride;
        FN(rbd_rnea)(m, in, in + n, qdd ? qdd + (size_t)k * n : NULL, gravity, c_out + (size_t)k * n, v, a, f);
    }
}

/* dc_du[k*2n^2 + col*n + row] from q_qd[k*stride + ...] (+ qdd[k*n+i] if qdd != NULL, else qdd = 0) */
void FN(rbd_rnea_grad_batch)(const model_t *m, int N, const real *q_qd, int stride, const real *qdd, real gravity, real *dc_du_dev) {
    int n = m->n;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int k = 0; k < N; k++) {
        real out[2 * RBD_MAX_N * RBD_MAX_N], zero[RBD_MAX_N];
        for (int i = 0; i < n; i++) zero[i] = 0;
        const real *in = q_qd + (size_t)k * stride;
        FN(rbd_rnea_grad)(m, in, in + n, qdd ? qdd + (size_t)k * n : zero, gravity, out);
        real *dst = dc_du_dev + (size_t)k * 2 * n * n;
        for (int col = 0; col < 2 * n; col++) for (int r = 0; r < n; r++) dst[col * n + r] = out[r * 2 * n + col];
    }
}

/* Minv[k*n^2 + col*n + row], upper triangle (zeros below the diagonal) like the reference's direct_minv kernel */
void FN(rbd_minv_batch)(const model_t *m, int N, const real *q, int stride, real *Minv_dev) {
    int n = m->n;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int k = 0; k < N; k++) {
        real out[RBD_MAX_N * RBD_MAX_N];
        FN(rbd_minv)(m, q + (size_t)k * stride, out, 0);
        real *dst = Minv_dev + (size_t)k * n * n;
        for (int col = 0; col < n; col++) for (int r = 0; r < n; r++) dst[col * n + r] = out[r * n + col];
    }
}
