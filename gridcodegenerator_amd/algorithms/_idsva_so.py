"""Second-order derivatives of inverse dynamics (IDSVA-SO), emitter for the HIP/CDNA4 backend - serial revolute chains.

Mirrors the role of the reference's algorithms/_idsva_so.py (gen_idsva_so_inner :36-915, device :925, kernel :958, host :1030): the four
n x n x n tensors  d2tau_dq2 | d2tau_dqd2 | d2tau_dvdq | dM_dq  (reference :204-208; index [i][j][k] = i*n*n + j*n + k) of
Singh, Russell & Wensing's second-order inverse-dynamics derivatives.  The reference works in the world frame with dense 6x6 tensors per
joint (crm/crf/icrf matrices, D1..D4, outer products t1..t9, p1..p6, :473-911) and ~30 block-wide phases.  Here:

  * all per-joint quantities come from the tip-frame machinery of algorithms/_tip_frame_gradient.py (S, Pd = psid, Pdd = psidd, the
    composites I^C (10 parameters), B^C (12), f^C, T1..T4); every bilinear form x^T D y is frame invariant, so nothing else changes;
  * lane c owns subtree joint c of every (joint j, ancestor an, subtree joint c) triple of the reference (:573-911) and never forms its
    D matrices: D1 y = S x* (I^C y) - I^C (S x y), D4 y = y x* T1, D3 = D1 + D4,
    D2 y = Pd x* (I^C y) + y x* (I^C Pd) - I^C (Pd x y) + S x* (B^C y) - B^C (S x y) are applied as operators to the wave-uniform
    vectors S_m, Pd_m, Pdd_m of one joint m at a time (read from one 20-float LDS record per joint);
  * the reference's "=" / "+=" phases are merged into ONE complete value per output entry (checked entry by entry against the literal
    restatement oracle/idsva_so_oracle.py by the tests), so every entry is written exactly once and no read-modify-write is needed.  The
    structurally zero entries of dM_dq (dM_ik/dq_j with j <= min(i, k)) are written by their owner lane first.  The kernels stage the
    4 n^3 record in LDS behind a compact slice and copy it out with coalesced 16-byte stores: 106 us per 16384 solves, of which 31 us
    arithmetic (LDS capacity allows 0.75 waves per SIMD).  Storing straight to global memory is store-issue bound however the entries are
    grouped (measured): all as 4-byte stores 145 us; the rows whose first index is the lane's joint collected into 16-byte pieces 121 us -
    every store instruction still moves at most 28 bytes per solve.  The device function writes through whatever pointer it is given.

Parity: the reference ships no oracle or vectors for this algorithm (PARITY UNPINNED); the tests compare with the NumPy restatement of
the reference's emitter, which is itself anchored on finite differences of the pinned first-order oracle.
Scope this round: single serial chains of revolute joints (self.tip_frame and one chain); other robots do not get the idsva_so surface.
"""
from ._tip_frame_gradient import _chain_step, _emit_bias, _emit_chain_decls, _emit_link_constants_load, _emit_link_setup


def gen_idsva_so_available(self):
    return bool(getattr(self, "tip_frame", False) and self.tip_nseg == 1)


def gen_idsva_so_inner_temp_mem_size(self):
    return 0  # one 20-value record per joint inside the scratch area (it re-uses the X(q) block)


def gen_idsva_so_lds_layout(self):
    """LDS of the idsva_so kernels: a compact per-solve slice [q | qd | qdd (padded) | scratch = X(q) / per-joint records (20 n) | qdd zeros] and,
    behind the block's slices, the 4 n^3 output tensors of every solve (staged so that the record leaves with coalesced 16-byte stores:
    scattered 4-byte global stores were measured at 4.7x the cost of the arithmetic).  Returns (slice, scratch, staging, threads)."""
    n, G = self.model.n, self.lanes_per_solve
    pad4 = lambda x: (x + 3) // 4 * 4
    scratch = 20 * n + pad4(n)
    sl = pad4(3 * n) + scratch
    if (sl // 4) % 2 == 0:
        sl += 4
    stage = 4 * n * n * n
    threads = 64
    while threads > G and (threads // G) * (sl + stage) * 4 > 150 * 1024:
        threads -= G
    return sl, scratch, stage, threads


def gen_idsva_so_inner_function_call(self, use_thread_group=False, use_qdd_input=False, updated_var_names=None):
    self.gen_add_code_line("idsva_so_inner<T>(so, s_qd, s_qdd, s_X, d_robotModel, gravity, lane, active);")


def gen_idsva_so_inner(self, use_thread_group=False, use_qdd_input=False):
    n = self.model.n
    n2, n3 = n * n, n * n * n
    self.gen_add_func_doc("Computes the second order derivatives of inverse dynamics",
                          ["serial revolute chains; computed in the frame of the tip link, lane c owns the subtree joint of every (joint, ancestor, subtree) triple",
                           "so = [d2tau_dq2 | d2tau_dqd2 | d2tau_dvdq | dM_dq], each n x n x n with [i][j][k] at i*n*n + j*n + k (reference algorithms/_idsva_so.py:204-208);",
                           "d2tau_dvdq[i][j][k] = d2 tau_i / dq_j dqd_k, dM_dq[i][j][k] = d M_ik / dq_j.  Every entry is written exactly once; `so` may be global or LDS memory"],
                          ["so is the output record of this solve (4*NUM_JOINTS^3 values)", "s_qd is the vector of joint velocities in LDS",
                           "s_qdd is the vector of joint accelerations in LDS", "s_X is this solve's compact X(q) storage; it is overwritten by the per-joint records [S | Pd | Pdd]",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU", "gravity is the gravity constant",
                           "lane is the caller's lane index inside the solve's lane group", "active is false for lane groups without a solve (they compute but do not store)"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void idsva_so_inner(T *__restrict__ so, const T *__restrict__ s_qd, const T *__restrict__ s_qdd, T *__restrict__ s_X, const robotModel<T> *d_robotModel, const T gravity, const int lane, const bool active) {", True)
    _emit_link_constants_load(self)
    _emit_chain_decls(self)
    for i in range(self.tip_L - 1, -1, -1):
        _chain_step(self, i)
    self.gen_add_sync(use_thread_group)  # every lane is done with s_X before the records overwrite it
    _emit_link_setup(self)
    self.gen_add_code_line("const T qdd = (lane < %d) ? s_qdd[lane] : static_cast<T>(0);" % n)
    _emit_bias(self, True)
    lines = """
T Pdd[6]; grid_mxm(Pdd, a, S); grid_mxm_peq(Pdd, v, Pd);
// per-lane vectors of the reference's backward pass (:508-526): T1 = I^C S, T2 = -(B^C)^T S (bottom half zero), T3, T4, and I^C Pd
T T1[6], T2[3], T3[6], T4[6], ICPd[6];
grid_rbi_mul(T1, IC, S);
{ T t[3]; grid_btmul(t, BC, S); T2[0] = -t[0]; T2[1] = -t[1]; T2[2] = -t[2]; }
grid_rbi_mul(ICPd, IC, Pd);
grid_bmul(T3, BC, Pd); grid_rbi_mul_peq(T3, IC, Pdd, static_cast<T>(1)); grid_fxv_peq(T3, S, fC);
grid_bmul(T4, BC, S);
#pragma unroll
for (int r = 0; r < 6; r++) { T4[r] += static_cast<T>(2)*ICPd[r]; }
if (lane < @N@) { // record of joint `lane`: [S | Pd | Pdd]
    T *rec = &s_X[20*lane];
    #pragma unroll
    for (int r = 0; r < 6; r++) { rec[r] = S[r]; rec[6 + r] = Pd[r]; rec[12 + r] = Pdd[r]; }
}
grid_wave_sync();
const bool own = active && (lane < @N@)@NOSTORE@;
T *q2 = so, *qd2 = so + @N3@, *vq = so + 2*@N3@, *mq = so + 3*@N3@;
const int c = lane;
// structurally zero entries of dM_dq: dM_ik/dq_j with j <= min(i, k); the owner lane (largest index) writes them first
#pragma unroll 1
for (int b = 0; b < @N@; b++) {
    for (int e = 0; e <= b; e++) {
        if (own && b <= c) { mq[(c*@N@ + e)*@N@ + b] = static_cast<T>(0); mq[(b*@N@ + e)*@N@ + c] = static_cast<T>(0); }
    }
}
// main loops: joint m supplies the vectors this lane's operators act on, joint l the vector the results are dotted with
#pragma unroll 1
for (int m = 0; m < @N@; m++) {
    T yS[6], yP[6], yPP[6];
    #pragma unroll
    for (int r = 0; r < 6; r++) { yS[r] = s_X[20*m + r]; yP[r] = s_X[20*m + 6 + r]; yPP[r] = s_X[20*m + 12 + r]; }
    T d1S[6], d1P[6], d1PP[6], d2S[6], d2P[6], d3S[6], d3P[6], d4S[6];
    {
        T IyS[6], IyP[6], sxS[6], sxP[6], t[6], u[6], w[6];
        grid_rbi_mul(IyS, IC, yS); grid_rbi_mul(IyP, IC, yP);
        grid_mxm(sxS, S, yS); grid_mxm(sxP, S, yP);
        // D1 y = S x* (I^C y) - I^C (S x y)
        grid_rbi_mul(t, IC, sxS);
        #pragma unroll
        for (int r = 0; r < 6; r++) { d1S[r] = -t[r]; }
        grid_fxv_peq(d1S, S, IyS);
        grid_rbi_mul(t, IC, sxP);
        #pragma unroll
        for (int r = 0; r < 6; r++) { d1P[r] = -t[r]; }
        grid_fxv_peq(d1P, S, IyP);
        grid_mxm(u, S, yPP); grid_rbi_mul(t, IC, u); grid_rbi_mul(w, IC, yPP);
        #pragma unroll
        for (int r = 0; r < 6; r++) { d1PP[r] = -t[r]; }
        grid_fxv_peq(d1PP, S, w);
        // D4 y = y x* T1 ; D3 = D1 + D4
        grid_zero6(d4S); grid_fxv_peq(d4S, yS, T1);
        #pragma unroll
        for (int r = 0; r < 6; r++) { d3S[r] = d1S[r] + d4S[r]; d3P[r] = d1P[r]; }
        grid_fxv_peq(d3P, yP, T1);
        // D2 y = Pd x* (I^C y) + y x* (I^C Pd) - I^C (Pd x y) + S x* (B^C y) - B^C (S x y)
        grid_mxm(t, Pd, yS); grid_rbi_mul(u, IC, t); grid_bmul(w, BC, sxS);
        #pragma unroll
        for (int r = 0; r < 6; r++) { d2S[r] = -u[r] - w[r]; }
        grid_fxv_peq(d2S, Pd, IyS); grid_fxv_peq(d2S, yS, ICPd); grid_bmul(t, BC, yS); grid_fxv_peq(d2S, S, t);
        grid_mxm(t, Pd, yP); grid_rbi_mul(u, IC, t); grid_bmul(w, BC, sxP);
        #pragma unroll
        for (int r = 0; r < 6; r++) { d2P[r] = -u[r] - w[r]; }
        grid_fxv_peq(d2P, Pd, IyP); grid_fxv_peq(d2P, yP, ICPd); grid_bmul(t, BC, yP); grid_fxv_peq(d2P, S, t);
    }
    #pragma unroll @UNROLL_L@
    for (int l = 0; l < @N@; l++) {
        T xS[6], xP[6];
        #pragma unroll
        for (int r = 0; r < 6; r++) { xS[r] = s_X[20*l + r]; xP[r] = s_X[20*l + 6 + r]; }
        const T x_d1S = grid_dot6(xS, d1S), x_d3S = grid_dot6(xS, d3S), x_d2S = grid_dot6(xS, d2S), x_d4S = grid_dot6(xS, d4S);
        const T x_d1P = grid_dot6(xS, d1P), x_d3P = grid_dot6(xS, d3P), x_d2P = grid_dot6(xS, d2P), x_d1PP = grid_dot6(xS, d1PP);
        if (l >= m) { // joint j = l, ancestor an = m (reference phases t1..t5 and the p1/p2 terms, :566-710, :876-886)
            const int j = l, an = m;
            const bool ok = own && (c >= j), ne = ok && (c != j);
            T p1[6], p2[6]; grid_mxm(p1, yP, xS); grid_mxm(p2, yPP, xS);
            const T pa = -(p1[0]*T2[0] + p1[1]*T2[1] + p1[2]*T2[2]) + grid_dot6(p2, T1);
            const T vq_ = -grid_dot6(xP, d3P) + pa;
            const T w_ = x_d2P + x_d1PP;
            if (ok) {
                q2[(c*@N@ + an)*@N@ + j] = vq_;
                vq[(c*@N@ + an)*@N@ + j] = -x_d3P;
                if (an < j) { q2[(c*@N@ + j)*@N@ + an] = vq_; qd2[(c*@N@ + j)*@N@ + an] = -x_d3S; qd2[(c*@N@ + an)*@N@ + j] = -x_d3S; }
                else { qd2[(c*@N@ + an)*@N@ + j] = -x_d1S; }
            }
            if (ne) {
                q2[(j*@N@ + c)*@N@ + an] = w_; q2[(j*@N@ + an)*@N@ + c] = w_;
                vq[(j*@N@ + an)*@N@ + c] = x_d3P;
                qd2[(j*@N@ + c)*@N@ + an] = x_d3S; qd2[(j*@N@ + an)*@N@ + c] = x_d3S;
                vq[(j*@N@ + c)*@N@ + an] = x_d2S + static_cast<T>(2)*x_d1P;
            }
        }
        if (l <= m) { // joint j = m, ancestor an = l: second half of the reference's t8 phase (:815-816)
            const int j = m, an = l;
            if (own && (c > j)) { mq[(an*@N@ + c)*@N@ + j] = x_d1S; mq[(j*@N@ + c)*@N@ + an] = x_d1S; }
        }
        if (l < m) { // joint j = m, ancestor an = l < j (reference phases t6..t9, the p3..p6 terms, :725-911)
            const int j = m, an = l;
            const bool ok = own && (c >= j), ne = ok && (c != j);
            T p1[6], p3[6], p4[6]; grid_mxm(p1, xP, yS); grid_mxm(p3, xS, yS); grid_mxm(p4, yP, xS);
            #pragma unroll
            for (int r = 0; r < 6; r++) { p4[r] = static_cast<T>(2)*(p1[r] - p4[r]); }
            const T pb = -(p3[0]*T2[0] + p3[1]*T2[1] + p3[2]*T2[2]) + grid_dot6(p4, T1);
            const T pc3 = -grid_dot6(p3, T3), pc4 = -grid_dot6(p3, T4); // p5 = S_j x S_an = -p3
            const T w_ = x_d2P + x_d1PP;
            if (ok) {
                vq[(c*@N@ + j)*@N@ + an] = -x_d3P + pb;
                q2[(an*@N@ + j)*@N@ + c] = w_ - pc3;
                vq[(an*@N@ + j)*@N@ + c] = x_d3P - pc4;
                mq[(an*@N@ + j)*@N@ + c] = x_d4S; mq[(c*@N@ + j)*@N@ + an] = x_d4S;
            }
            if (ne) {
                q2[(an*@N@ + c)*@N@ + j] = w_ - pc3;
                qd2[(an*@N@ + j)*@N@ + c] = x_d3S; qd2[(an*@N@ + c)*@N@ + j] = x_d3S;
                vq[(an*@N@ + c)*@N@ + j] = x_d2S + static_cast<T>(2)*x_d1P;
            }
            if (ok && (c == j)) { // d2tau_dqd2[an][j][j] = p6 . S_j with the lane's own T1 = I^C_j S_j (:901-911)
                T t[6]; grid_zero6(t); grid_fxv_peq(t, S, T1);
                qd2[(an*@N@ + j)*@N@ + j] = grid_dot6(T1, p3) + grid_dot6(xS, t);
            }
        }
    }
}
""".replace("@N@", str(n)).replace("@N3@", str(n3))
    # timing ablation only (GRID_DEBUG_STOP=30): everything is computed, (almost) nothing is stored
    lines = lines.replace("@UNROLL_L@", str(self.tuning["so_unroll"] or n))  # tuning knob: inner-loop unrolling of idsva_so (full: -7 % vs none on the 7-DoF arm)
    lines = lines.replace("@NOSTORE@", " && (gravity < static_cast<T>(-1e30))" if self.tuning["debug_stop"] == 30 else "")
    for line in lines.strip("\n").split("\n"):
        self.gen_add_code_line(line)
    self.gen_add_end_function()


def gen_idsva_so_device(self, use_thread_group=False, use_qdd_input=False):
    n = self.model.n
    params = ["so is the output record of this solve: 4*NUM_JOINTS^3 values [d2tau_dq2 | d2tau_dqd2 | d2tau_dvdq | dM_dq] (global or LDS memory)",
              "s_q is the vector of joint positions in LDS", "s_qd is the vector of joint velocities in LDS"]
    if use_qdd_input:
        params.append("s_qdd is the vector of joint accelerations in LDS")
    params += ["s_scratch is LDS scratch of IDSVA_SO_SCRATCH_PER_SOLVE elements (X(q), then the per-joint records)", "d_robotModel is the pointer to the initialized model specific helpers on the GPU",
               "gravity is the gravity constant", "lane is the caller's lane index inside the solve's lane group", "active is false for lane groups without a solve"]
    self.gen_add_func_doc("Computes the second order derivatives of inverse dynamics: X(q) update + idsva_so_inner (lane-group cooperative)",
                          ["all lanes of the solve's lane group must call it" + ("" if use_qdd_input else "; qdd = 0")], params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void idsva_so_device(T *so, const T *s_q, const T *s_qd, " + ("const T *s_qdd, " if use_qdd_input else "") +
                           "T *s_scratch, const robotModel<T> *d_robotModel, const T gravity, const int lane, const bool active) {", True)
    self.gen_add_code_line("T *s_X = s_scratch;")
    if not use_qdd_input:
        self.gen_add_code_line("T *s_qdd = &s_scratch[%d];" % (20 * n))
        self.gen_add_code_line("if (lane < %d) { s_qdd[lane] = static_cast<T>(0); }" % n)
    self.gen_load_update_XImats_helpers_function_call(use_thread_group)
    self.gen_idsva_so_inner_function_call(use_thread_group, use_qdd_input)
    self.gen_add_end_function()


def gen_idsva_so_kernel(self, use_thread_group=False, use_qdd_input=False, single_call_timing=False):
    n = self.model.n
    n3 = n * n * n
    func_params = ["d_idsva_so is the output: 4*NUM_JOINTS^3 values per solve, [d2tau_dq2 | d2tau_dqd2 | d2tau_dvdq | dM_dq]",
                   "d_q_qd_u is the vector of joint positions, velocities (and torques, unused)", "stride_q_qd_u is the stride between each q, qd, u",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)"]
    func_def = "void idsva_so_kernel(T *d_idsva_so, const T *d_q_qd_u, const int stride_q_qd_u, "
    if use_qdd_input:
        func_def += "const T *d_qdd, "
        func_params.insert(-3, "d_qdd is the vector of joint accelerations")
    func_def += "const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {"
    if single_call_timing:
        func_def = func_def.replace("kernel(", "kernel_single_timing(")
    self.gen_add_func_doc("Computes the second order derivatives of inverse dynamics",
                          ["launch with IDSVA_SO_SUGGESTED_THREADS threads and IDSVA_SO_DYNAMIC_SHARED_MEM_COUNT*sizeof(T) bytes of dynamic LDS: the 4 n^3 record of every solve is staged in LDS"], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__global__ GRID_LAUNCH_BOUNDS")
    self.gen_add_code_line(func_def, True)
    sl, scratch, stage, threads = self.gen_idsva_so_lds_layout()
    pad3n = (3 * n + 3) // 4 * 4
    self.gen_kernel_prologue("IDSVA_SO_LDS_PER_SOLVE", "IDSVA_SO_MAX_SOLVES_PER_BLOCK")
    self.gen_add_code_line("T *s_q_qd_u = s_mem; T *s_q = s_q_qd_u; T *s_qd = &s_q_qd_u[%d]; T *s_qdd = &s_q_qd_u[%d]; T *s_scratch = &s_mem[%d]; (void)s_qdd;" % (n, 2 * n, pad3n))
    self.gen_add_code_line("T *s_idsva_so = &s_out_all[grp*%d]; // this solve's output record, staged in LDS" % stage)
    if single_call_timing:
        self.gen_add_code_line("const int k = 0; const int kc = 0; const bool valid = (blockIdx.x + blockIdx.y == 0) && (grp == 0); const int lane = lane_id;")
        self.gen_add_code_line("if (!valid) {return;}")
    else:
        self.gen_add_parallel_loop("k", "NUM_TIMESTEPS", use_thread_group, block_level=True)
    if use_qdd_input:
        self.gen_kernel_load_inputs("q_qd_u", "stride_q_qd_u", 2 * n, use_thread_group, "qdd", n, n)
    else:
        self.gen_kernel_load_inputs("q_qd_u", "stride_q_qd_u", 2 * n, use_thread_group)
    if single_call_timing:
        self.gen_add_code_line("for (int rep = 0; rep < NUM_TIMESTEPS; rep++){", True)
    self.gen_add_code_line("// compute")
    self.gen_add_code_line("idsva_so_device<T>(s_idsva_so, s_q, s_qd, " + ("s_qdd, " if use_qdd_input else "") + "s_scratch, d_robotModel, gravity, lane, true);")
    if single_call_timing:
        self.gen_add_end_control_flow()
    if single_call_timing:
        self.gen_kernel_save_result_single_timing("idsva_so", 4 * n3, use_thread_group)
    else:
        self.gen_kernel_save_result("idsva_so", 4 * n3, 4 * n3, use_thread_group)
    if not single_call_timing:
        self.gen_add_end_control_flow()
    self.gen_add_end_function()


def gen_idsva_so_host(self, mode=0):
    n = self.model.n
    single_call_timing = mode == 1
    compute_only = mode == 2
    func_params = ["hd_data is the packaged input and output pointers",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant,",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)",
                   "streams are pointers to HIP streams for async memory transfers (if needed)"]
    name = "idsva_so_host" + ("_single_timing" if single_call_timing else "") + ("_compute_only" if compute_only else "")
    self.gen_add_func_doc("Computes the second order derivatives of inverse dynamics", [], func_params, None)
    self.gen_add_code_line("template <typename T, bool USE_QDD_FLAG = false>")
    self.gen_add_code_line("__host__")
    self.gen_add_code_line("void " + name + "(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps,")
    self.gen_add_code_line("                      const dim3 block_dimms, const dim3 thread_dimms" + ("" if compute_only else ", hipStream_t *streams") + ") {", True)
    self.gen_add_code_line("int stride_q_qd = 3*NUM_JOINTS;")
    cnt = "" if single_call_timing else "num_timesteps*"
    if not compute_only:
        self.gen_add_code_lines(["// start code with memory transfer",
                                 "gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q_qd*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[0]));",
                                 "if (USE_QDD_FLAG) {gpuErrchk(hipMemcpyAsync(hd_data->d_qdd,hd_data->h_qdd,NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[1]));}",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    kern = "idsva_so_kernel" + ("_single_timing" if single_call_timing else "") + "<T>"
    self.gen_add_code_line("// then call the kernel")
    if single_call_timing:
        self.gen_add_code_line("struct timespec start, end; clock_gettime(CLOCK_MONOTONIC,&start);")
    self.gen_add_code_lines(["if (USE_QDD_FLAG) {hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, IDSVA_SO_LDS_PER_SOLVE, IDSVA_SO_STAGE_PER_SOLVE, IDSVA_SO_MAX_SOLVES_PER_BLOCK),0,hd_data->d_idsva_so,hd_data->d_q_qd_u,stride_q_qd,hd_data->d_qdd,d_robotModel,gravity,num_timesteps);}",
                             "else {hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, IDSVA_SO_LDS_PER_SOLVE, IDSVA_SO_STAGE_PER_SOLVE, IDSVA_SO_MAX_SOLVES_PER_BLOCK),0,hd_data->d_idsva_so,hd_data->d_q_qd_u,stride_q_qd,d_robotModel,gravity,num_timesteps);}",
                             "gpuErrchk(hipGetLastError()); gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("clock_gettime(CLOCK_MONOTONIC,&end);")
    if not compute_only:
        self.gen_add_code_lines(["// finally transfer the result back",
                                 "gpuErrchk(hipMemcpy(hd_data->h_idsva_so,hd_data->d_idsva_so,4*NUM_JOINTS*NUM_JOINTS*NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyDeviceToHost));",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("printf(\"Single Call IDSVA_SO %fus\\n\",time_delta_us_timespec(start,end)/static_cast<double>(num_timesteps));")
    self.gen_add_end_function()


def gen_idsva_so(self, use_thread_group=False):
    if not self.gen_idsva_so_available():
        return
    self.gen_idsva_so_inner(use_thread_group, True)
    self.gen_idsva_so_device(use_thread_group, False)
    self.gen_idsva_so_device(use_thread_group, True)
    for qdd in (True, False):
        for timing in (True, False):
            self.gen_idsva_so_kernel(use_thread_group, qdd, timing)
    for mode in (0, 1, 2):
        self.gen_idsva_so_host(mode)
