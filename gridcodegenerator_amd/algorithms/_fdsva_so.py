"""Second-order derivatives of forward dynamics (FDSVA-SO), emitter for the HIP/CDNA4 backend - serial revolute chains.

Mirrors the role of the reference's algorithms/_fdsva_so.py (gen_fdsva_so_inner :3-85, device :121-157, kernel :159-230, host :232-300):
df2 = [d2a_dqdq | d2a_dvdv | d2a_dvdq | d2a_dtdq], four n x n x n tensors (flat index i*n*n + j*n + k) with
    d2a_dqdq[i][j][k] = d2 qdd_i / dq_j dq_k,   d2a_dvdv[i][j][k] = d2 qdd_i / dqd_j dqd_k,
    d2a_dvdq[i][j][k] = d2 qdd_i / dq_j dqd_k,  d2a_dtdq[i][j][k] = d Minv_ik / dq_j.
The device function composes what the reference composes (:147-156): forward dynamics (qdd), its gradient (df/du), M^-1, idsva_so at that
qdd, then the contraction of :52-81:
    inner_dq[L][k][j]    = sum_P dM[L][j][P] dfdq[P][k] + sum_P dM[L][k][P] dfdq[P][j] + d2tau_dqdq[L][k][j]
    inner_cross[L][k][j] = sum_P dM[L][k][P] dfdqd[P][j] + d2tau_dvdq[L][k][j]
    inner_tau[L][k][j]   = sum_P dM[L][k][P] Minv[j][P]
    out_x[i][k][j]       = -sum_L Minv[L][i] inner_x[L][k][j]          (x = dq, cross, tau; d2tau_dvdv itself for d2a_dvdv)
Lane j owns the output entries (., k, j): it keeps column j of df/dq, df/dqd and row j of M^-1 in registers, walks k and L, and stores
runs that are contiguous along j.  The idsva_so tensors of the solve stay in LDS (4 n^3 values), so a block carries fewer solves than the
first-order kernels (FDSVA_SO_SUGGESTED_THREADS).

Parity: PARITY UNPINNED (the reference ships no oracle or vectors); oracle/fdsva_so_oracle.py restates the reference's contraction and is
anchored on finite differences of the pinned first-order forward-dynamics-gradient oracle.
Scope: every robot idsva_so is emitted for (gen_idsva_so_mode: serial revolute chains and kinematic trees of revolute joints; large robots keep the
idsva_so tensors in a global workspace, gen_idsva_so_direct).
"""


# contraction for large robots: the register form below keeps 7 n values per lane and unrolls n^2 bodies; here only the 4 n output accumulators of one
# (k, j) stay in registers, L and p are loops and the operands are re-read (LDS / L2) - n^4 multiply-adds either way
_ROLLED = """
const T *tqq = s_idsva_so, *tvv = s_idsva_so + @N3@, *tvq = s_idsva_so + 2*@N3@, *dM = s_idsva_so + 3*@N3@;
const int j = (lane < @N@) ? lane : 0;
const bool own = active && (lane < @N@);
const T *fq_j = &s_df_du[j*@N@], *fv_j = &s_df_du[(@N@ + j)*@N@], *mi_j = &s_Minv[j*@LD@]; // column j of df/dq and of df/dqd, row j of M^-1
#pragma unroll 1
for (int k = 0; k < @N@; k++) {
    const T *fq_k = &s_df_du[k*@N@];
    T oq[@N@], oc[@N@], ov[@N@], ot[@N@]; // out_x[i][k][j], i = 0..n-1
    #pragma unroll
    for (int i = 0; i < @N@; i++) { oq[i] = oc[i] = ov[i] = ot[i] = static_cast<T>(0); }
    #pragma unroll 1
    for (int L = 0; L < @N@; L++) {
        const T *dMk = &dM[(L*@N@ + k)*@N@], *dMj = &dM[(L*@N@ + j)*@N@];
        T aq = tqq[(L*@N@ + k)*@N@ + j], ac = tvq[(L*@N@ + k)*@N@ + j], at = static_cast<T>(0);
        const T av = tvv[(L*@N@ + k)*@N@ + j];
        #pragma unroll 6
        for (int p = 0; p < @N@; p++) { const T mk = dMk[p]; aq += dMj[p]*fq_k[p] + mk*fq_j[p]; ac += mk*fv_j[p]; at += mk*mi_j[p]; }
        #pragma unroll
        for (int i = 0; i < @N@; i++) { const T mi = s_Minv[L*@LD@ + i]; oq[i] += mi*aq; oc[i] += mi*ac; ov[i] += mi*av; ot[i] += mi*at; }
    }
    if (own) {
        #pragma unroll
        for (int i = 0; i < @N@; i++) {
            const int e = (i*@N@ + k)*@N@ + j;
            df2[e] = -oq[i]; df2[@N3@ + e] = -ov[i]; df2[2*@N3@ + e] = -oc[i]; df2[3*@N3@ + e] = -ot[i];
        }
    }
}
"""


# the same contraction reading the idsva_so tensors from the COMPACT staging record (algorithms/_idsva_so.py: gen_idsva_so_compact_layout): symmetric entries
# through their canonical index, dM_dq[L][a][p] = d M_Lp / d q_a from mqc[tri(max(L, p)) + min(L, p)][a], structurally zero for a <= min(L, p)
_COMPACT = """
const T *q2c = s_idsva_so + @Q2@, *qd2c = s_idsva_so + @QD2@, *tvq = s_idsva_so + @VQ@, *mqc = s_idsva_so + @MQ@;
@SPLIT@
T fq_j[@N@], fv_j[@N@], mi_j[@N@]; // column j of df/dq and of df/dqd, row j of M^-1
#pragma unroll
for (int p = 0; p < @N@; p++) { fq_j[p] = s_df_du[j*@N@ + p]; fv_j[p] = s_df_du[(@N@ + j)*@N@ + p]; mi_j[p] = s_Minv[j*@LD@ + p]; }
#pragma unroll 1
for (int k = @K0@; k < @N@; k += @KSTEP@) {
    T fq_k[@N@];
    #pragma unroll
    for (int p = 0; p < @N@; p++) { fq_k[p] = s_df_du[k*@N@ + p]; }
    const int hi = (k > j) ? k : j, lo = (k > j) ? j : k, kj = (hi*(hi + 1) >> 1) + lo; // (k, j) in the symmetric tensors
    T rq[@N@], rc[@N@], rv[@N@], rt[@N@]; // inner_dq, inner_cross, d2tau_dvdv, inner_tau at (L, k, j), L = 0..n-1
    #pragma unroll
    for (int L = 0; L < @N@; L++) {
        T aq = q2c[L*@TRI@ + kj], ac = tvq[(L*@N@ + k)*@N@ + j], at = static_cast<T>(0);
        #pragma unroll
        for (int p = 0; p < @N@; p++) {
            const int plo = (L < p) ? L : p, phi = (L < p) ? p : L, base = ((phi*(phi + 1) >> 1) + plo)*@N@; // (compile-time after unrolling)
            const T mk_ = mqc[base + k], mj_ = mqc[base + j]; // (slots of structural zeros are never written: read, then discarded by the select - no exec-masked load blocks)
            const T mk = (k > plo) ? mk_ : static_cast<T>(0), mj = (j > plo) ? mj_ : static_cast<T>(0);
            aq += mj*fq_k[p] + mk*fq_j[p]; ac += mk*fv_j[p]; at += mk*mi_j[p];
        }
        rq[L] = aq; rc[L] = ac; rt[L] = at; rv[L] = qd2c[L*@TRI@ + kj];
    }
    #pragma unroll
    for (int i = 0; i < @N@; i++) {
        T oq = static_cast<T>(0), oc = static_cast<T>(0), ov = static_cast<T>(0), ot = static_cast<T>(0);
        #pragma unroll
        for (int L = 0; L < @N@; L++) { const T mi = s_Minv[L*@LD@ + i]; oq += mi*rq[L]; oc += mi*rc[L]; ov += mi*rv[L]; ot += mi*rt[L]; }
        if (own) {
            const int e = (i*@N@ + k)*@N@ + j;
            df2[e] = -oq; df2[@N3@ + e] = -ov; df2[2*@N3@ + e] = -oc; df2[3*@N3@ + e] = -ot;
        }
    }
}
"""


# _COMPACT with the results of ALL k of a lane held in registers and stored at the end, (i, tensor) by (i, tensor): the lane group then writes the 4 n^2-byte run of an
# (i, tensor) within a few instructions.  With the k loop outside, one store instruction wrote 8 n bytes of every run and came back to it an iteration (~1.5 k
# instructions) later: 45 MB of half-written lines in flight at 65 536 solves, more than the L2 holds - HBM write traffic 2.0x the result (profiles/r03_so_iiwa14_pmc.txt)
_COMPACT_HELD_HEAD = """
const T *q2c = s_idsva_so + @Q2@, *qd2c = s_idsva_so + @QD2@, *tvq = s_idsva_so + @VQ@, *mqc = s_idsva_so + @MQ@;
@SPLIT@
T fq_j[@N@], fv_j[@N@], mi_j[@N@]; // column j of df/dq and of df/dqd, row j of M^-1
#pragma unroll
for (int p = 0; p < @N@; p++) { fq_j[p] = s_df_du[j*@N@ + p]; fv_j[p] = s_df_du[(@N@ + j)*@N@ + p]; mi_j[p] = s_Minv[j*@LD@ + p]; }
"""
_COMPACT_HELD_K = """
T oq@KK@[@N@], oc@KK@[@N@], ov@KK@[@N@], ot@KK@[@N@]; // the results (i, k, j) of this lane's k = @K0@ + @KSTEP@*@KK@ (sign flipped at the store)
{
    const int kx = @K0@ + @KSTEP@*@KK@; const int k = (kx < @N@) ? kx : 0; // (a lane whose last k is past the end computes k = 0 again and stores nothing)
    T fq_k[@N@];
    #pragma unroll
    for (int p = 0; p < @N@; p++) { fq_k[p] = s_df_du[k*@N@ + p]; }
    const int hi = (k > j) ? k : j, lo = (k > j) ? j : k, kj = (hi*(hi + 1) >> 1) + lo; // (k, j) in the symmetric tensors
    #pragma unroll
    for (int i = 0; i < @N@; i++) { oq@KK@[i] = oc@KK@[i] = ov@KK@[i] = ot@KK@[i] = static_cast<T>(0); }
    #pragma unroll 1
    for (int L = 0; L < @N@; L++) { // inner_dq, inner_cross, d2tau_dvdv, inner_tau at (L, k, j), then their share of every row i
        T aq = q2c[L*@TRI@ + kj], ac = tvq[(L*@N@ + k)*@N@ + j], at = static_cast<T>(0);
        const T av = qd2c[L*@TRI@ + kj];
        #pragma unroll
        for (int p = 0; p < @N@; p++) {
            const int plo = (L < p) ? L : p, phi = (L < p) ? p : L, base = ((phi*(phi + 1) >> 1) + plo)*@N@; // (compile-time after unrolling)
            const T mk_ = mqc[base + k], mj_ = mqc[base + j]; // (slots of structural zeros are never written: read, then discarded by the select)
            const T mk = (k > plo) ? mk_ : static_cast<T>(0), mj = (j > plo) ? mj_ : static_cast<T>(0);
            aq += mj*fq_k[p] + mk*fq_j[p]; ac += mk*fv_j[p]; at += mk*mi_j[p];
        }
        #pragma unroll
        for (int i = 0; i < @N@; i++) { const T mi = s_Minv[L*@LD@ + i]; oq@KK@[i] += mi*aq; oc@KK@[i] += mi*ac; ov@KK@[i] += mi*av; ot@KK@[i] += mi*at; }
    }
    asm volatile("" ::: "memory"); // (the next k re-reads its operands from LDS: values kept live across the bodies would not fit the register file)
}
"""


def _compact_held(kt, hold):
    """The held form for kt values of k per lane: a rolled loop over groups of `hold` ADJACENT k (lane (j, kh) of a group takes k = hold*KSTEP*t + hold*kh + 0..hold-1), the body
    once per k of the group (distinct register arrays), then the group's stores (i, tensor) by (i, tensor): hold*KSTEP*n values of a run at once."""
    text = _COMPACT_HELD_HEAD + "#pragma unroll 1\nfor (int t = 0; t < %d; t++) {\n" % (-(-kt // hold))
    text += "".join(_COMPACT_HELD_K.replace("@K0@ + @KSTEP@*@KK@", "%d*@KSTEP@*t + %d*@K0@ + @KK@" % (hold, hold)).replace("@KK@", str(kk)) for kk in range(hold))
    for kk in range(hold):  # (one predicated block per k: the group's lanes write adjacent k, the blocks follow each other within a few dozen instructions)
        text += "{ const int k = %d*@KSTEP@*t + %d*@K0@ + %d;\n  if (own && k < @N@) {\n    T *dst = &df2[k*@N@ + j];\n    #pragma unroll\n    for (int i = 0; i < @N@; i++) { dst[i*@N@*@N@] = -oq%d[i]; dst[@N3@ + i*@N@*@N@] = -ov%d[i]; dst[2*@N3@ + i*@N@*@N@] = -oc%d[i]; dst[3*@N3@ + i*@N@*@N@] = -ot%d[i]; }\n  } }\n" % (hold, hold, kk, kk, kk, kk, kk)
    return text + "}\n"



# robots with several base-rooted components (a quadruped's legs; the humanoid's torso tree and legs): a fixed base decouples them, so qdd_i and all
# its derivatives involve only the joints of the component of joint i - M^-1, df/du and the idsva_so tensors are block diagonal over the components
# (contiguous joint ranges in DFS pre-order) and so is the result.  Every loop of the contraction runs over the component of the lane's joint
# (uniform trip count = the largest component, shorter components masked): 18^4 + 2 6^4 instead of 30^4 multiply-adds per tensor on the 30-DoF
# humanoid (7.5x fewer), 4 3^4 instead of 12^4 on the quadruped; the entries that couple different components are exact zeros (zero fill of the record).
_BLOCKED = """
@SPLIT@
const int c0 = grid_so_component[2*j], cs = grid_so_component[2*j + 1]; // first joint and size of the base-rooted component of joint j
@TENSORS@
if (active) { for (int e = lane; e < 4*@N3@; e += GRID_LANES_PER_SOLVE) { df2[e] = static_cast<T>(0); } } // (entries that couple different components; the same wave overwrites the others below)
grid_wave_sync(); // (the fill and the entries are ordered by the wave's program order; lane-per-thread execution models need the hand-off point)
@COLJ@
#pragma unroll 1
for (int kk = @K0@; kk < @MAXC@; kk += @KSTEP@) {
    const bool vk = kk < cs; const int k = c0 + (vk ? kk : 0);
    const T *fq_k = &s_df_du[k*@N@ + c0];
    T oq[@MAXC@], oc[@MAXC@], ov[@MAXC@], ot[@MAXC@]; // out_x[c0 + i][k][j]
    #pragma unroll
    for (int i = 0; i < @MAXC@; i++) { oq[i] = oc[i] = ov[i] = ot[i] = static_cast<T>(0); }
    #pragma unroll 1
    for (int LL = 0; LL < @MAXC@; LL++) {
        const bool vL = LL < cs; const int L = c0 + (vL ? LL : 0);
        @ROWS@
        #pragma unroll @PUNROLL@
        for (int pp = 0; pp < @MAXC@; pp++) {
            const bool vp = pp < cs; const int p = vp ? pp : 0;
            const T mk = vp ? dMk[p] : static_cast<T>(0), mj = vp ? dMj[p] : static_cast<T>(0);
            aq += mj*fq_k[p] + mk*fq_j[@PJ@]; ac += mk*fv_j[@PJ@]; at += mk*mi_j[@PJ@];
        }
        #pragma unroll
        for (int i = 0; i < @MAXC@; i++) { const T mi = (vL && i < cs) ? s_Minv[L*@LD@ + c0 + ((i < cs) ? i : 0)] : static_cast<T>(0); oq[i] += mi*aq; oc[i] += mi*ac; ov[i] += mi*av; ot[i] += mi*at; }
    }
    if (own && vk) {
        #pragma unroll
        for (int i = 0; i < @MAXC@; i++) {
            if (i < cs) {
                const int e = ((c0 + i)*@N@ + k)*@N@ + j;
                df2[e] = -oq[i]; df2[@N3@ + e] = -ov[i]; df2[2*@N3@ + e] = -oc[i]; df2[3*@N3@ + e] = -ot[i];
            }
        }
    }
}
"""


def gen_fdsva_so_components(self):
    """[first joint, size] of the base-rooted component of every joint (flat, one pair per lane of the lane group), and the largest size; None for robots with a single
    component (or where the compact staging of the chain form is used)."""
    m = self.model
    if len(m.roots) < 2 or self.gen_idsva_so_compact() or not self.tuning["so_blocked"]:
        return None
    comp = {}
    for r in m.roots:
        for j_ in m.subtree[r]:
            comp[j_] = (r, len(m.subtree[r]))
    flat = []
    for j_ in range(self.lanes_per_solve):
        flat += list(comp.get(j_, (0, 0)))
    return flat, max(len(m.subtree[r]) for r in m.roots)


def gen_fdsva_so_inner_temp_mem_size(self):
    return 0


def gen_fdsva_so_stage_size(self):
    """Per-solve staging behind the block's slices: df/du (2 n^2, padded) then the idsva_so tensors (4 n^3; in the direct form of large robots they
    stay in a global workspace, gen_idsva_so_direct)."""
    n = self.model.n
    return (2 * n * n + 3) // 4 * 4 + (0 if self.gen_idsva_so_direct() else (self.gen_idsva_so_compact_layout()["SIZE"] if self.gen_idsva_so_packed() else 4 * n * n * n))


def gen_fdsva_so_inner(self, use_thread_group=False):
    n = self.model.n
    n2, n3 = n * n, n * n * n
    ld = self.minv_ld
    self.gen_add_func_doc("Second Order of Forward Dynamics with Spatial Vector Algebra: the contraction of the idsva_so tensors with M^-1 and df/du",
                          ["lane j produces the entries (., k, j) of the four output tensors and stores each exactly once; all lanes of the lane group must call it"],
                          ["df2 is the output record of this solve: 4*NUM_JOINTS^3 values [d2a_dqdq | d2a_dvdv | d2a_dvdq | d2a_dtdq] (global or LDS memory)",
                           "s_idsva_so are the second derivative tensors of inverse dynamics at qdd = FD(q, qd, u), in LDS (large robots: global workspace)" + ("; COMPACT staging record (idsva_so_inner_compact)" if self.gen_idsva_so_packed() else ""),
                           "s_Minv is the dense symmetric inverse mass matrix in LDS (leading dimension GRID_MINV_LD)",
                           "s_df_du is the gradient of the forward dynamics in LDS ([col*n + row], col in [0, 2n))",
                           "lane is the caller's lane index inside the solve's lane group", "active is false for lane groups without a solve"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void fdsva_so_inner(T *df2, const T *s_idsva_so, const T *s_Minv, const T *s_df_du, const int lane, const bool active) {", True)
    compact = self.gen_idsva_so_compact()
    blocked = self.gen_fdsva_so_components()
    G0 = self.lanes_per_solve
    kstep = 2 if ((n <= 12 or compact) and G0 // 2 >= n) else 1
    kt = -(-n // kstep)  # k values per lane
    hold = min(int(self.tuning["so_hold"]), kt)
    held = compact and hold > 0 and 4 * n * hold <= 140 and int(self.tuning["debug_stop"]) != 31
    lines = ((_compact_held(kt, hold) if held else _COMPACT) if compact else _BLOCKED if blocked is not None else _ROLLED if n > 12 else """
const T *tqq = s_idsva_so, *tvv = s_idsva_so + @N3@, *tvq = s_idsva_so + 2*@N3@, *dM = s_idsva_so + 3*@N3@;
@SPLIT@
T fq_j[@N@], fv_j[@N@], mi_j[@N@]; // column j of df/dq and of df/dqd, row j of M^-1
#pragma unroll
for (int p = 0; p < @N@; p++) { fq_j[p] = s_df_du[j*@N@ + p]; fv_j[p] = s_df_du[(@N@ + j)*@N@ + p]; mi_j[p] = s_Minv[j*@LD@ + p]; }
#pragma unroll 1
for (int k = @K0@; k < @N@; k += @KSTEP@) {
    T fq_k[@N@];
    #pragma unroll
    for (int p = 0; p < @N@; p++) { fq_k[p] = s_df_du[k*@N@ + p]; }
    T rq[@N@], rc[@N@], rv[@N@], rt[@N@]; // inner_dq, inner_cross, d2tau_dvdv, inner_tau at (L, k, j), L = 0..n-1
    #pragma unroll
    for (int L = 0; L < @N@; L++) {
        const T *dMk = &dM[(L*@N@ + k)*@N@], *dMj = &dM[(L*@N@ + j)*@N@];
        T aq = tqq[(L*@N@ + k)*@N@ + j], ac = tvq[(L*@N@ + k)*@N@ + j], at = static_cast<T>(0);
        #pragma unroll
        for (int p = 0; p < @N@; p++) { const T mk = dMk[p]; aq += dMj[p]*fq_k[p] + mk*fq_j[p]; ac += mk*fv_j[p]; at += mk*mi_j[p]; }
        rq[L] = aq; rc[L] = ac; rt[L] = at; rv[L] = tvv[(L*@N@ + k)*@N@ + j];
    }
    #pragma unroll
    for (int i = 0; i < @N@; i++) {
        T oq = static_cast<T>(0), oc = static_cast<T>(0), ov = static_cast<T>(0), ot = static_cast<T>(0);
        #pragma unroll
        for (int L = 0; L < @N@; L++) { const T mi = s_Minv[L*@LD@ + i]; oq += mi*rq[L]; oc += mi*rc[L]; ov += mi*rv[L]; ot += mi*rt[L]; }
        if (own) {
            const int e = (i*@N@ + k)*@N@ + j;
            df2[e] = -oq; df2[@N3@ + e] = -ov; df2[2*@N3@ + e] = -oc; df2[3*@N3@ + e] = -ot;
        }
    }
}
""")
    if compact and int(self.tuning["debug_stop"]) == 31:  # timing ablation (wrong results): only one of the four result tensors is stored
        lines = lines.replace("df2[e] = -oq; df2[@N3@ + e] = -ov; df2[2*@N3@ + e] = -oc; df2[3*@N3@ + e] = -ot;",
                              "df2[e] = -oq; if (df2 == nullptr) { df2[@N3@ + e] = -ov; df2[2*@N3@ + e] = -oc; df2[3*@N3@ + e] = -ot; }")
    G = self.lanes_per_solve
    if (n <= 12 or compact) and G // 2 >= n:  # lane groups at least twice as wide as the robot has joints (the `wide` instances): the two halves of the group take alternate k
        lines = lines.replace("@SPLIT@", "const int jh = lane %% %d, kh = lane / %d; // joint and k parity of this lane\nconst int j = (jh < @N@) ? jh : 0;\nconst bool own = active && (jh < @N@);" % (G // 2, G // 2)).replace("@K0@", "kh").replace("@KSTEP@", "2")
    else:
        lines = lines.replace("@SPLIT@", "const int j = (lane < @N@) ? lane : 0;\nconst bool own = active && (lane < @N@);").replace("@K0@", "0").replace("@KSTEP@", "1")
    if blocked is not None:
        if self.gen_idsva_so_blocks():  # the tensors are staged as the dense blocks of the components: [tensor][L - c0][k - c0][j - c0] at the block's base
            tensors = ("const int cs3 = cs*cs*cs; const T *tqq = s_idsva_so + grid_so_blocks[3*j + 2], *tvv = tqq + cs3, *tvq = tqq + 2*cs3, *dM = tqq + 3*cs3; const int jl = j - c0; // (block staging)")
            rows = ("const int LLc = vL ? LL : 0, kkc = vk ? kk : 0; const T *dMk = &dM[(LLc*cs + kkc)*cs], *dMj = &dM[(LLc*cs + jl)*cs];\n"
                    "        T aq = tqq[(LLc*cs + kkc)*cs + jl], ac = tvq[(LLc*cs + kkc)*cs + jl], at = static_cast<T>(0);\n"
                    "        const T av = tvv[(LLc*cs + kkc)*cs + jl];")
        else:
            tensors = "const T *tqq = s_idsva_so, *tvv = s_idsva_so + @N3@, *tvq = s_idsva_so + 2*@N3@, *dM = s_idsva_so + 3*@N3@;"
            rows = ("const T *dMk = &dM[(L*@N@ + k)*@N@ + c0], *dMj = &dM[(L*@N@ + j)*@N@ + c0];\n"
                    "        T aq = tqq[(L*@N@ + k)*@N@ + j], ac = tvq[(L*@N@ + k)*@N@ + j], at = static_cast<T>(0);\n"
                    "        const T av = tvv[(L*@N@ + k)*@N@ + j];")
        if blocked[1] <= 12:  # small components: column j of df/dq, df/dqd and row j of M^-1 wait in registers (three LDS reads fewer per inner step)
            colj = ("T fq_j[@MAXC@], fv_j[@MAXC@], mi_j[@MAXC@]; // column j of df/dq and of df/dqd, row j of M^-1 (rows of the component)\n"
                    "#pragma unroll\n"
                    "for (int p = 0; p < @MAXC@; p++) { const int pc = (p < cs) ? p : 0; fq_j[p] = s_df_du[j*@N@ + c0 + pc]; fv_j[p] = s_df_du[(@N@ + j)*@N@ + c0 + pc]; mi_j[p] = s_Minv[j*@LD@ + c0 + pc]; }")
        else:
            colj = "const T *fq_j = &s_df_du[j*@N@ + c0], *fv_j = &s_df_du[(@N@ + j)*@N@ + c0], *mi_j = &s_Minv[j*@LD@ + c0]; // column j of df/dq and of df/dqd, row j of M^-1 (rows of the component)"
        lines = lines.replace("@COLJ@", colj).replace("@PJ@", "pp" if blocked[1] <= 12 else "p")
        lines = lines.replace("@TENSORS@", tensors).replace("@ROWS@", rows)
        lines = lines.replace("@PUNROLL@", str(blocked[1] if blocked[1] <= 12 else 6)).replace("@MAXC@", str(blocked[1]))
    if compact:
        L_ = self.gen_idsva_so_compact_layout()
        for k_ in ("Q2", "QD2", "VQ", "MQ", "TRI"):
            lines = lines.replace("@%s@" % k_, str(L_[k_]))
    lines = lines.replace("@N3@", str(n3)).replace("@N@", str(n)).replace("@LD@", str(ld))
    for line in lines.strip("\n").split("\n"):
        self.gen_add_code_line(line)
    self.gen_add_end_function()


def gen_fdsva_so_split(self):
    """True where the host wrappers and the C ABI run fdsva_so as TWO kernels (records beyond the LDS of a CU, several base-rooted components, largest component
    <= 22 joints - the 30-DoF humanoid): fdsva_so_prepare_kernel (lane groups: gradient, M^-1, idsva_so into the global workspace) and
    fdsva_so_contract_kernel (one 256-thread block per solve: thread <-> entry (k, j) of a component, the dM_dq slabs staged in LDS).  The single
    fdsva_so_kernel stays for callers that launch it themselves.  Returns (pairs per thread, largest component) or None."""
    if not (self.gen_idsva_so_direct() and self.tuning["so_split"]):
        return None
    comps = self.gen_fdsva_so_components()
    if comps is None:
        return None
    maxc = comps[1]
    npairs = -(-maxc * maxc // 256)
    if npairs * 4 * maxc > 160:
        return None
    return npairs, maxc


def gen_fdsva_so_device(self, use_thread_group=False, prepare=False):
    n = self.model.n
    if prepare:
        self.gen_add_func_doc("First half of fdsva_so (lane-group cooperative): forward dynamics gradient, M^-1 and idsva_so at the solution; the contraction is a kernel of its own (fdsva_so_contract_kernel)",
                              ["all lanes of the solve's lane group must call it; s_df_du holds the first-order gradient and &s_work[GRID_OFF_MINV] the dense M^-1 on return"],
                              ["s_df_du is a pointer to LDS for the derivative of forward dynamics WRT q,qd of size 2*NUM_JOINTS*NUM_JOINTS = " + str(2 * n * n),
                               "s_idsva_so is this solve's record in the global workspace for the 4*NUM_JOINTS^3 second derivative tensors of inverse dynamics",
                               "s_q is the vector of joint positions", "s_qd is the vector of joint velocities", "s_u is the vector of joint control inputs",
                               "s_work is this solve's LDS workspace of GRID_LDS_PER_SOLVE elements",
                               "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                               "gravity is the gravity constant", "lane is the caller's lane index inside the solve's lane group", "active is false for lane groups without a solve"], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__device__ __forceinline__")
        self.gen_add_code_line("void fdsva_so_prepare_device(T *s_df_du, T *s_idsva_so, const T *s_q, const T *s_qd, const T *s_u, T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane, const bool active) {", True)
        self.gen_add_code_line("T *s_qdd = &s_work[GRID_OFF_QDD]; T *s_Minv = &s_work[GRID_OFF_MINV];")
        assert getattr(self, "branch_frame", False) or not self.tip_frame
        if getattr(self, "branch_frame", False):
            self.gen_add_code_line("forward_dynamics_gradient_device<T>(s_df_du, s_q, s_qd, s_u, s_work, d_robotModel, gravity, lane);")
            self.gen_add_code_line("s_qdd = &s_work[FD_DU_OFF_QDD];")
        else:
            self.gen_add_code_line("forward_dynamics_gradient_device<T>(s_df_du, s_q, s_qd, s_u, s_work, d_robotModel, gravity, lane);")
            self.gen_add_code_line("forward_dynamics_device<T>(s_qdd, s_q, s_qd, s_u, s_work, d_robotModel, gravity, lane);")
        self.gen_add_code_line("direct_minv_device<T>(s_Minv, s_q, s_work, d_robotModel, lane);")
        self.gen_add_code_line("idsva_so_device<T>(s_idsva_so, s_q, s_qd, s_qdd, &s_work[GRID_OFF_X], d_robotModel, gravity, lane, active, true); // (only the component blocks of the workspace are zero-filled: what the contraction reads)")
        self.gen_add_sync(use_thread_group)
        self.gen_add_end_function()
        return
    self.gen_add_func_doc("Second Order of Forward Dynamics with Spatial Vector Algebra (lane-group cooperative): forward dynamics, its gradient, M^-1, idsva_so at the solution, contraction",
                          ["all lanes of the solve's lane group must call it; s_df_du holds the first-order gradient on return (reference algorithms/_fdsva_so.py:147-156)"],
                          ["df2 is the output record of this solve: 4*NUM_JOINTS^3 values (global or LDS memory)",
                           "s_df_du is a pointer to LDS for the derivative of forward dynamics WRT q,qd of size 2*NUM_JOINTS*NUM_JOINTS = " + str(2 * n * n),
                           "s_idsva_so is a pointer to LDS (large robots: to a global workspace) for the 4*NUM_JOINTS^3 second derivative tensors of inverse dynamics",
                           "s_q is the vector of joint positions", "s_qd is the vector of joint velocities", "s_u is the vector of joint control inputs",
                           "s_work is this solve's LDS workspace of GRID_LDS_PER_SOLVE elements",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                           "gravity is the gravity constant", "lane is the caller's lane index inside the solve's lane group", "active is false for lane groups without a solve"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void fdsva_so_device(T *df2, T *s_df_du, T *s_idsva_so, const T *s_q, const T *s_qd, const T *s_u, T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane, const bool active) {", True)
    self.gen_add_code_line("T *s_qdd = &s_work[GRID_OFF_QDD]; T *s_Minv = &s_work[GRID_OFF_MINV];")
    self.gen_add_code_line("// (the gradient comes first: it uses the workspace freely - the M^-1 slot as scratch, on branched robots its own layout - so qdd and M^-1 are produced after it)")
    if self.gen_tip_frame_fused_so():
        self.gen_add_code_line("// (serial chains: the tip-frame inner of the gradient leaves qdd and M^-1 behind and goes straight on to the idsva_so main loops with the per-joint")
        self.gen_add_code_line("//  quantities it holds in registers - one pass over frames, velocities and composites where the reference composes four calls, algorithms/_fdsva_so.py:147-156)")
        self.gen_add_code_line("T *s_X = &s_work[GRID_OFF_X]; T *s_U = &s_work[GRID_OFF_U];")
        self.gen_load_update_XImats_helpers_function_call(use_thread_group)
        self.gen_add_code_line("forward_dynamics_gradient_inner_tip_so<T>(s_df_du, s_qd, s_u, s_X, s_U, s_Minv, d_robotModel, gravity, lane, s_qdd, s_Minv, s_idsva_so, s_X, %s);" % ("active" if self.gen_idsva_so_direct() else "true"))
        self.gen_add_sync(use_thread_group)
        self.gen_add_code_line("fdsva_so_inner<T>(df2, s_idsva_so, s_Minv, s_df_du, lane, active);")
        self.gen_add_end_function()
        return
    if self.tip_frame and not getattr(self, "branch_frame", False):
        self.gen_add_code_line("// (robots whose gradient runs the tip-frame inner: it leaves qdd and M^-1 behind - one pass instead of three)")
        self.gen_add_code_line("forward_dynamics_gradient_device<T>(s_df_du, s_q, s_qd, s_u, s_work, d_robotModel, gravity, lane, s_qdd, s_Minv);")
    elif getattr(self, "branch_frame", False):
        self.gen_add_code_line("// (branch-frame robots: the gradient's inner leaves qdd in its slice, FD_DU_OFF_QDD - inside the input block, which nothing below overwrites)")
        self.gen_add_code_line("forward_dynamics_gradient_device<T>(s_df_du, s_q, s_qd, s_u, s_work, d_robotModel, gravity, lane);")
        self.gen_add_code_line("s_qdd = &s_work[FD_DU_OFF_QDD];")
        self.gen_add_code_line("direct_minv_device<T>(s_Minv, s_q, s_work, d_robotModel, lane);")
    else:
        self.gen_add_code_line("forward_dynamics_gradient_device<T>(s_df_du, s_q, s_qd, s_u, s_work, d_robotModel, gravity, lane);")
        self.gen_add_code_line("forward_dynamics_device<T>(s_qdd, s_q, s_qd, s_u, s_work, d_robotModel, gravity, lane);")
        self.gen_add_code_line("direct_minv_device<T>(s_Minv, s_q, s_work, d_robotModel, lane);")
    self.gen_add_code_line("idsva_so_device%s<T>(s_idsva_so, s_q, s_qd, s_qdd, &s_work[GRID_OFF_X], d_robotModel, gravity, lane, %s);" % ("_compact" if self.gen_idsva_so_packed() else "", "active" if self.gen_idsva_so_direct() else "true"))
    self.gen_add_sync(use_thread_group)
    self.gen_add_code_line("fdsva_so_inner<T>(df2, s_idsva_so, s_Minv, s_df_du, lane, active);")
    self.gen_add_end_function()


def gen_fdsva_so_fused_layout(self):
    """LDS of the fdsva_so KERNELS where the fused inner runs (gen_tip_frame_fused_so, records staged in LDS).  What is live when the idsva_so record is
    written and contracted - the per-joint records [S | Pd | Pdd], df/du and M^-1 - makes up the slice; everything the gradient works with before that
    (inputs, X(q), its hand-off records, M) lives INSIDE the staging record of the idsva_so tensors, which is not written until the gradient is done.
    7-DoF arm: 300 + 936 values per solve instead of 372 + 1 036 -> 8 resident waves per CU instead of 7.  None where the kernels keep the general slice."""
    if not self.gen_tip_frame_fused_so() or self.gen_idsva_so_direct():
        return None
    n = self.model.n
    pad4 = lambda x: (x + 3) // 4 * 4
    stage = self.gen_idsva_so_compact_layout()["SIZE"] if self.gen_idsva_so_compact() else 4 * n ** 3
    L = {"REC": 0, "DF_DU": 20 * n, "MINV": 20 * n + pad4(2 * n * n)}
    sl = L["MINV"] + n * self.minv_ld
    if (sl // 4) % 2 == 0:
        sl += 4
    L["SLICE"], L["STAGE"] = sl, stage
    L["IN"] = 0
    L["X"] = pad4(3 * n)
    L["G"] = L["X"] + 20 * n
    L["M"] = L["G"] + pad4(self.tip_rec * n)
    if L["M"] + n * self.minv_ld > stage:
        return None
    return L


def gen_fdsva_so_lds_per_solve(self):
    """(slice, staging) values per solve of the fdsva_so kernels."""
    F = self.gen_fdsva_so_fused_layout()
    if F is not None:
        return F["SLICE"], F["STAGE"]
    return self.gen_lds_layout()["TOTAL"], self.gen_fdsva_so_stage_size()


def gen_fdsva_so_fused_device(self, use_thread_group=False):
    """The kernels' form of fdsva_so_device with the compact LDS layout of gen_fdsva_so_fused_layout."""
    n = self.model.n
    F = self.gen_fdsva_so_fused_layout()
    self.gen_add_func_doc("Second Order of Forward Dynamics with Spatial Vector Algebra (lane-group cooperative): what the fdsva_so kernels run",
                          ["as fdsva_so_device, with the kernels' own LDS layout: the gradient works inside the staging record of the idsva_so tensors (not written until it is done),",
                           "the slice holds only what is live afterwards: per-joint records, df/du, M^-1; all lanes of the solve's lane group must call it"],
                          ["df2 is the output record of this solve: 4*NUM_JOINTS^3 values (global or LDS memory)",
                           "s_slice is this solve's LDS slice of FDSVA_SO_LDS_PER_SOLVE elements",
                           "s_stage is this solve's staging record of FDSVA_SO_STAGE_PER_SOLVE elements; q | qd | u are in its first 3*NUM_JOINTS elements on entry",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                           "gravity is the gravity constant", "lane is the caller's lane index inside the solve's lane group", "active is false for lane groups without a solve"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void fdsva_so_fused_device(T *df2, T *s_slice, T *s_stage, const robotModel<T> *d_robotModel, const T gravity, const int lane, const bool active) {", True)
    self.gen_add_code_line("const T *s_q = &s_stage[%d]; const T *s_qd = &s_stage[%d]; const T *s_u = &s_stage[%d];" % (F["IN"], F["IN"] + n, F["IN"] + 2 * n))
    self.gen_add_code_line("T *s_X = &s_stage[%d]; T *s_G = &s_stage[%d]; T *s_M = &s_stage[%d]; // the gradient's working set, inside the (not yet written) idsva_so record" % (F["X"], F["G"], F["M"]))
    self.gen_add_code_line("T *s_rec = &s_slice[%d]; T *s_df_du = &s_slice[%d]; T *s_Minv = &s_slice[%d];" % (F["REC"], F["DF_DU"], F["MINV"]))
    self.gen_load_update_XImats_helpers_function_call(use_thread_group)
    self.gen_add_code_line("forward_dynamics_gradient_inner_tip_so<T>(s_df_du, s_qd, s_u, s_X, s_G, s_M, d_robotModel, gravity, lane, static_cast<T *>(nullptr), s_Minv, s_stage, s_rec, true);")
    self.gen_add_sync(use_thread_group)
    self.gen_add_code_line("fdsva_so_inner<T>(df2, s_stage, s_Minv, s_df_du, lane, active);")
    self.gen_add_end_function()


def gen_fdsva_so_kernel(self, use_thread_group=False, single_call_timing=False):
    n = self.model.n
    n3 = n * n * n
    stage = self.gen_fdsva_so_stage_size()
    F = self.gen_fdsva_so_fused_layout()
    func_params = ["d_df2 is the output: 4*NUM_JOINTS^3 values per solve, [d2a_dqdq | d2a_dvdv | d2a_dvdq | d2a_dtdq]",
                   "d_q_qd_u is the vector of joint positions, velocities, and input torques", "stride_q_qd_u is the stride between each q, qd, u",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)"]
    direct = self.gen_idsva_so_direct()
    func_def = "void fdsva_so_kernel(T *d_df2, " + ("T *d_idsva_so, " if direct else "") + "const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {"
    if direct:
        func_params.insert(1, "d_idsva_so is a workspace of 4*NUM_JOINTS^3 values per solve (the idsva_so tensors of a solve do not fit LDS; gridData::d_idsva_so): it holds them on return")
    if single_call_timing:
        func_def = func_def.replace("kernel(", "kernel_single_timing(")
    self.gen_add_func_doc("Second Order of Forward Dynamics with Spatial Vector Algebra",
                          ["launch with FDSVA_SO_SUGGESTED_THREADS threads and FDSVA_SO_DYNAMIC_SHARED_MEM_COUNT*sizeof(T) bytes of dynamic LDS: every solve keeps its 4 n^3 idsva_so tensors in LDS"],
                          func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__global__ GRID_LAUNCH_BOUNDS")
    self.gen_add_code_line(func_def, True)
    self.gen_kernel_prologue("FDSVA_SO_LDS_PER_SOLVE", "FDSVA_SO_MAX_SOLVES_PER_BLOCK")
    if F is not None:
        self.gen_add_code_lines(["T *s_stage = &s_out_all[grp*%d]; // staging record of the idsva_so tensors; until they are written it holds the inputs and the gradient's working set" % F["STAGE"],
                                 "T *s_q_qd_u = &s_stage[%d];" % F["IN"]])
    else:
        self.gen_add_code_lines(["T *s_q_qd_u = &s_mem[GRID_OFF_IN]; T *s_q = s_q_qd_u; T *s_qd = &s_q_qd_u[%d]; T *s_u = &s_q_qd_u[%d];" % (n, 2 * n),
                                 "T *s_df_du = &s_out_all[grp*%d];" % stage + ("" if direct else " T *s_idsva_so = s_df_du + %d;" % ((2 * n * n + 3) // 4 * 4))])
    if single_call_timing:
        self.gen_add_code_line("const int k = 0; const int kc = 0; const bool valid = (blockIdx.x + blockIdx.y == 0) && (grp == 0); const int lane = lane_id;")
        self.gen_add_code_line("if (!valid) {return;}")
    else:
        self.gen_add_parallel_loop("k", "NUM_TIMESTEPS", use_thread_group, block_level=True)
    self.gen_kernel_load_inputs("q_qd_u", "stride_q_qd_u", 3 * n, use_thread_group)
    if single_call_timing:
        self.gen_add_code_line("for (int rep = 0; rep < NUM_TIMESTEPS; rep++){", True)
    self.gen_add_code_line("// compute; the record of solve k goes straight to global memory")
    if direct:
        self.gen_add_code_line("T *s_idsva_so = &d_idsva_so[static_cast<size_t>(kc)*%d]; // (global workspace; lane groups without a solve never write through it)" % (4 * n3))
    if F is not None:
        self.gen_add_code_line("fdsva_so_fused_device<T>(&d_df2[static_cast<size_t>(kc)*%d], s_mem, s_stage, d_robotModel, gravity, lane, valid);" % (4 * n3))
    else:
        self.gen_add_code_line("fdsva_so_device<T>(&d_df2[static_cast<size_t>(kc)*%d], s_df_du, s_idsva_so, s_q, s_qd, s_u, s_mem, d_robotModel, gravity, lane, valid);" % (4 * n3))
    self.gen_add_sync(use_thread_group)
    if single_call_timing:
        self.gen_add_end_control_flow()
    if not single_call_timing:
        self.gen_add_end_control_flow()
    self.gen_add_end_function()


def gen_fdsva_so_split_kernels(self, use_thread_group=False):
    """fdsva_so_prepare_kernel + fdsva_so_contract_kernel (see gen_fdsva_so_split)."""
    n = self.model.n
    n2, n3 = n * n, n * n * n
    npairs, maxc = self.gen_fdsva_so_split()
    ld = self.minv_ld
    stage = self.gen_fdsva_so_stage_size()
    A = self.gen_add_code_line
    self.gen_add_func_doc("First kernel of the two-kernel form of fdsva_so: forward dynamics gradient, M^-1 and the idsva_so tensors at qdd = FD(q, qd, u), all to global memory",
                          ["launch like fdsva_so_kernel (FDSVA_SO_SUGGESTED_THREADS threads, FDSVA_SO_DYNAMIC_SHARED_MEM_COUNT*sizeof(T) bytes of dynamic LDS); then fdsva_so_contract_kernel"],
                          ["d_idsva_so receives the 4*NUM_JOINTS^3 idsva_so tensors of every solve", "d_df_du receives df/du (2*NUM_JOINTS^2 values per solve, [col*n + row])",
                           "d_Minv receives the dense symmetric M^-1 (NUM_JOINTS^2 values per solve)",
                           "d_q_qd_u is the vector of joint positions, velocities, and input torques", "stride_q_qd_u is the stride between each q, qd, u",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)", "gravity is the gravity constant",
                           "num_timesteps is the length of the trajectory points we need to compute over"], None)
    A("template <typename T>")
    A("__global__ GRID_LAUNCH_BOUNDS")
    A("void fdsva_so_prepare_kernel(T *d_idsva_so, T *d_df_du, T *d_Minv, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {", True)
    self.gen_kernel_prologue("FDSVA_SO_LDS_PER_SOLVE", "FDSVA_SO_MAX_SOLVES_PER_BLOCK")
    self.gen_add_code_lines(["T *s_q_qd_u = &s_mem[GRID_OFF_IN]; T *s_q = s_q_qd_u; T *s_qd = &s_q_qd_u[%d]; T *s_u = &s_q_qd_u[%d];" % (n, 2 * n),
                             "T *s_df_du = &s_out_all[grp*%d];" % stage])
    self.gen_add_parallel_loop("k", "NUM_TIMESTEPS", use_thread_group, block_level=True)
    self.gen_kernel_load_inputs("q_qd_u", "stride_q_qd_u", 3 * n, use_thread_group)
    A("fdsva_so_prepare_device<T>(s_df_du, &d_idsva_so[static_cast<size_t>(kc)*%d], s_q, s_qd, s_u, s_mem, d_robotModel, gravity, lane, valid);" % (4 * n3))
    A("if (valid) { // df/du and the dense M^-1 of this solve: what the contraction kernel reads besides the tensors", True)
    A("const T *s_Minv = &s_mem[GRID_OFF_MINV];")
    A("for (int e = lane; e < %d; e += GRID_LANES_PER_SOLVE) { d_df_du[static_cast<size_t>(k)*%d + e] = s_df_du[e]; }" % (2 * n2, 2 * n2))
    A("for (int e = lane; e < %d; e += GRID_LANES_PER_SOLVE) { d_Minv[static_cast<size_t>(k)*%d + e] = s_Minv[(e / %d)*%d + (e %% %d)]; }" % (n2, n2, n, ld, n))
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)
    self.gen_add_end_control_flow()
    self.gen_add_end_function()

    comps = [(r, len(self.model.subtree[r])) for r in self.model.roots]
    C2 = maxc * maxc
    self.gen_add_func_doc("Second kernel of the two-kernel form of fdsva_so: the contraction of the idsva_so tensors with M^-1 and df/du (reference algorithms/_fdsva_so.py:52-81)",
                          ["one block of FDSVA_SO_CONTRACT_THREADS = 256 threads per solve (grid-stride over the batch), FDSVA_SO_CONTRACT_LDS*sizeof(T) bytes of dynamic LDS",
                           "block diagonal over the base-rooted components (a fixed base decouples them): thread <-> entry (k, j) of a component; per L the slab dM_dq[L] of the component",
                           "is staged in LDS, T1 = dM_dq[L] x [df/dq | df/dqd | M^-1] by every thread for its entry, the transposed term of inner_dq through LDS, and the entry's",
                           "four inner values go into 4 x (component size) accumulators out_x[i][k][j] = -sum_L Minv[L][i] inner_x[L][k][j]; the record is zero-filled first"],
                          ["d_df2 is the output: 4*NUM_JOINTS^3 values per solve, [d2a_dqdq | d2a_dvdv | d2a_dvdq | d2a_dtdq]",
                           "d_idsva_so, d_df_du, d_Minv are what fdsva_so_prepare_kernel left behind", "num_timesteps is the length of the trajectory points we need to compute over"], None)
    A("const int FDSVA_SO_CONTRACT_THREADS = 256;")
    A("const int FDSVA_SO_CONTRACT_LDS = %d; // df/dq | df/dqd | M^-1 | dM_dq slab | T1 of one component (%d x %d each)" % (5 * C2, maxc, maxc))
    A("template <typename T>")
    A("__global__ __launch_bounds__(256)")
    A("void fdsva_so_contract_kernel(T *d_df2, const T *d_idsva_so, const T *d_df_du, const T *d_Minv, const int NUM_TIMESTEPS) {", True)
    A("T *s_Fq = reinterpret_cast<T *>(grid_smem_raw), *s_Fv = s_Fq + %d, *s_Mi = s_Fv + %d, *s_dM = s_Mi + %d, *s_A = s_dM + %d;" % (C2, C2, C2, C2))
    A("const int tid = threadIdx.x + threadIdx.y*blockDim.x;")
    A("for (int k0 = blockIdx.x + blockIdx.y*gridDim.x; k0 < NUM_TIMESTEPS; k0 += gridDim.x*gridDim.y) {", True)
    A("const T *so = &d_idsva_so[static_cast<size_t>(k0)*%d]; const T *dfdu = &d_df_du[static_cast<size_t>(k0)*%d]; const T *Minv = &d_Minv[static_cast<size_t>(k0)*%d];" % (4 * n3, 2 * n2, n2))
    A("T *df2 = &d_df2[static_cast<size_t>(k0)*%d];" % (4 * n3))
    A("{ const T z4[4] = {static_cast<T>(0), static_cast<T>(0), static_cast<T>(0), static_cast<T>(0)}; // entries that couple different components are exact zeros")
    A("  for (int e = tid; e < %d; e += 256) { grid_store4(df2 + 4*e, z4); } }" % n3)
    A("__syncthreads(); // (the block's zero stores are complete before any thread overwrites an entry)")
    for (c0, cs) in comps:
        A("{ // base-rooted component: joints %d .. %d" % (c0, c0 + cs - 1), True)
        A("for (int e = tid; e < %d; e += 256) { const int p = e / %d, j = e %% %d; // [p][j]: df/dq, df/dqd and M^-1 of the component" % (cs * cs, cs, cs))
        A("  s_Fq[e] = dfdu[(%d + j)*%d + %d + p]; s_Fv[e] = dfdu[(%d + %d + j)*%d + %d + p]; s_Mi[e] = Minv[(%d + j)*%d + %d + p]; }" % (c0, n, c0, n, c0, n, c0, c0, n, c0))
        np_c = -(-cs * cs // 256)
        nsl = -(-cs * cs // 256)  # slab elements per thread
        A("#pragma unroll 1")
        A("for (int rnd = 0; rnd < %d; rnd++) { // entries (k, j) of the component, 256 per round (the accumulators of one entry per thread: %d registers)" % (np_c, 4 * cs), True)
        if np_c == 1:
            A("const bool ok = tid < %d; const int kk = ok ? tid / %d : 0, jj = ok ? tid %% %d : 0; (void)rnd;" % (cs * cs, cs, cs))
        else:
            # an entry and its transpose (k, j) <-> (j, k) must be in the SAME round (inner_dq takes T1[j][k] from the thread of the transposed entry):
            # round 0 = the 16 x 16 block k, j < 16, round 1 = the rest (rows k >= 16, then the columns j >= 16 of the rows k < 16)
            assert np_c == 2 and cs * cs - 256 <= 256
            A("bool ok; int kk, jj;")
            A("if (rnd == 0) { ok = true; kk = tid >> 4; jj = tid & 15; }")
            A("else if (tid < %d) { ok = true; kk = 16 + tid / %d; jj = tid %% %d; }" % ((cs - 16) * cs, cs, cs))
            A("else { const int e1 = tid - %d; ok = e1 < %d; kk = ok ? e1 / %d : 0; jj = ok ? 16 + e1 %% %d : 0; }" % ((cs - 16) * cs, 16 * (cs - 16), cs - 16, cs - 16))
        A("T oq[%d], ov[%d], oc[%d], ot[%d];" % (cs, cs, cs, cs))
        A("#pragma unroll")
        A("for (int i = 0; i < %d; i++) { oq[i] = ov[i] = oc[i] = ot[i] = static_cast<T>(0); }" % cs)
        A("// software pipeline: the slab dM_dq[L+1] and this entry's tensor values at L+1 are in flight while step L computes")
        A("T nsl[%d], nq, nc, nv;" % nsl)
        slab = lambda Lv: "; ".join("nsl[%d] = (tid + %d < %d) ? so[%d + ((%d + %s)*%d + %d + (tid + %d) / %d)*%d + %d + (tid + %d) %% %d] : static_cast<T>(0)" % (r, 256 * r, cs * cs, 3 * n3, c0, Lv, n, c0, 256 * r, cs, n, c0, 256 * r, cs) for r in range(nsl))
        vals = lambda Lv: "{ const int g = ((%d + %s)*%d + %d + kk)*%d + %d + jj; nq = so[g]; nc = so[%d + g]; nv = so[%d + g]; }" % (c0, Lv, n, c0, n, c0, 2 * n3, n3)
        A(slab("0") + ";")
        A(vals("0"))
        A("#pragma unroll 1")
        A("for (int L = 0; L < %d; L++) {" % cs, True)
        A("__syncthreads(); // (the previous slab and T1 are no longer read)")
        for r in range(nsl):
            A("if (tid + %d < %d) { s_dM[tid + %d] = nsl[%d]; } // dM_dq[L][k][p] of the component" % (256 * r, cs * cs, 256 * r, r))
        A("const T cq = nq, cc = nc, cv = nv;")
        A("__syncthreads();")
        A("{ const int Ln = (L + 1 < %d) ? L + 1 : L;" % cs)
        A("  " + slab("Ln") + ";")
        A("  " + vals("Ln") + " }")
        A("T aq = static_cast<T>(0), av = static_cast<T>(0), am = static_cast<T>(0);")
        A("#pragma unroll")
        A("for (int p = 0; p < %d; p++) { const T mk = s_dM[kk*%d + p]; aq += mk*s_Fq[p*%d + jj]; av += mk*s_Fv[p*%d + jj]; am += mk*s_Mi[p*%d + jj]; }" % (cs, cs, cs, cs, cs))
        A("if (ok) { s_A[kk*%d + jj] = aq; }" % cs)
        A("__syncthreads();")
        A("const T iq = aq + s_A[jj*%d + kk] + cq, ic = av + cc, it = am, iv = cv; // inner_dq, inner_cross, inner_tau, d2tau_dvdv at (L, k, j)" % cs)
        A("#pragma unroll")
        A("for (int i = 0; i < %d; i++) { const T mi = s_Mi[L*%d + i]; oq[i] += mi*iq; oc[i] += mi*ic; ov[i] += mi*iv; ot[i] += mi*it; }" % (cs, cs))
        self.gen_add_end_control_flow()
        A("if (ok) {", True)
        A("#pragma unroll")
        A("for (int i = 0; i < %d; i++) { const int e = ((%d + i)*%d + %d + kk)*%d + %d + jj; df2[e] = -oq[i]; df2[%d + e] = -ov[i]; df2[%d + e] = -oc[i]; df2[%d + e] = -ot[i]; }" % (cs, c0, n, c0, n, c0, n3, 2 * n3, 3 * n3))
        self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
        A("__syncthreads();")
        self.gen_add_end_control_flow()
    self.gen_add_end_control_flow()
    self.gen_add_end_function()


def gen_fdsva_so_host(self, mode=0):
    single_call_timing = mode == 1
    compute_only = mode == 2
    func_params = ["hd_data is the packaged input and output pointers",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant,",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)",
                   "streams are pointers to HIP streams for async memory transfers (if needed)"]
    name = "fdsva_so" + ("_single_timing" if single_call_timing else "") + ("_compute_only" if compute_only else "")
    self.gen_add_func_doc("Second Order of Forward Dynamics with Spatial Vector Algebra", ["thread_dimms must not exceed FDSVA_SO_SUGGESTED_THREADS threads"], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__host__")
    self.gen_add_code_line("void " + name + "(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps,")
    self.gen_add_code_line("                      const dim3 block_dimms, const dim3 thread_dimms" + ("" if compute_only else ", hipStream_t *streams") + ") {", True)
    if self.so_wide_lanes():
        self.gen_add_code_line("// (GRID_SO_WIDE) the kernels of the nested library for wider lane groups do the work; block_dimms may count lane groups of either width: the kernels grid-stride")
        self.gen_add_code_line("wide::" + name + "<T>(reinterpret_cast<wide::gridData<T> *>(hd_data), reinterpret_cast<const wide::robotModel<T> *>(d_robotModel), gravity, num_timesteps, block_dimms, thread_dimms" + ("" if compute_only else ", streams") + "); // (same layouts)")
        self.gen_add_end_function()
        return
    self.gen_add_code_line("int stride_q_qd_u = 3*NUM_JOINTS;")
    self.gen_add_code_line("if (num_timesteps > grid_so_max_timesteps<T>()) {gpuErrchk(hipErrorInvalidValue); return;} // (beyond what init_gridData allocates for second-order records)")
    cnt = "" if single_call_timing else "num_timesteps*"
    if not compute_only:
        self.gen_add_code_lines(["// start code with memory transfer",
                                 "gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q_qd_u*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[0]));",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    kern = "fdsva_so_kernel" + ("_single_timing" if single_call_timing else "") + "<T>"
    self.gen_add_code_line("// then call the kernel")
    if single_call_timing:
        self.gen_add_code_line("struct timespec start, end; clock_gettime(CLOCK_MONOTONIC,&start);")
    if self.gen_fdsva_so_split() is not None and not single_call_timing:
        self.gen_add_code_lines(["// two kernels (GRID_SO_SPLIT): gradient, M^-1 and the idsva_so tensors by lane groups, then the contraction with one block per solve",
                                 "hipLaunchKernelGGL((fdsva_so_prepare_kernel<T>),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, FDSVA_SO_LDS_PER_SOLVE, FDSVA_SO_STAGE_PER_SOLVE, FDSVA_SO_MAX_SOLVES_PER_BLOCK),0,hd_data->d_idsva_so,hd_data->d_df_du,hd_data->d_Minv,hd_data->d_q_qd_u,stride_q_qd_u,d_robotModel,gravity,num_timesteps);",
                                 "gpuErrchk(hipGetLastError());",
                                 "hipLaunchKernelGGL((fdsva_so_contract_kernel<T>),dim3(num_timesteps < 4096 ? (num_timesteps > 0 ? num_timesteps : 1) : 4096),dim3(FDSVA_SO_CONTRACT_THREADS),FDSVA_SO_CONTRACT_LDS*sizeof(T),0,hd_data->d_df2,hd_data->d_idsva_so,hd_data->d_df_du,hd_data->d_Minv,num_timesteps);",
                                 "gpuErrchk(hipGetLastError()); gpuErrchk(hipDeviceSynchronize());"])
    else:
        self.gen_add_code_lines(["hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, FDSVA_SO_LDS_PER_SOLVE, FDSVA_SO_STAGE_PER_SOLVE, FDSVA_SO_MAX_SOLVES_PER_BLOCK),0,hd_data->d_df2," + ("hd_data->d_idsva_so," if self.gen_idsva_so_direct() else "") + "hd_data->d_q_qd_u,stride_q_qd_u,d_robotModel,gravity,num_timesteps);",
                                 "gpuErrchk(hipGetLastError()); gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("clock_gettime(CLOCK_MONOTONIC,&end);")
    if not compute_only:
        self.gen_add_code_lines(["// finally transfer the result back",
                                 "gpuErrchk(hipMemcpy(hd_data->h_df2,hd_data->d_df2,4*NUM_JOINTS*NUM_JOINTS*NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyDeviceToHost));",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("printf(\"Single Call FDSVA_SO %fus\\n\",time_delta_us_timespec(start,end)/static_cast<double>(num_timesteps));")
    self.gen_add_end_function()


def gen_fdsva_so(self, use_thread_group=False):
    if not self.gen_idsva_so_available():
        return
    self.gen_fdsva_so_inner(use_thread_group)
    self.gen_fdsva_so_device(use_thread_group)
    if self.gen_fdsva_so_fused_layout() is not None:
        self.gen_fdsva_so_fused_device(use_thread_group)
    if self.gen_fdsva_so_split() is not None:
        self.gen_fdsva_so_device(use_thread_group, prepare=True)
        self.gen_fdsva_so_split_kernels(use_thread_group)
    self.gen_fdsva_so_kernel(use_thread_group, True)
    self.gen_fdsva_so_kernel(use_thread_group, False)
    for mode in (0, 1, 2):
        self.gen_fdsva_so_host(mode)
