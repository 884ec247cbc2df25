"""Analytical RNEA derivatives (dc/dq, dc/dqd), emitter for the HIP/CDNA4 backend.

Mirrors the role of the reference's algorithms/_inverse_dynamics_gradient.py (gen_inverse_dynamics_gradient_inner
:27-775, device :777, kernel :817, host :890) and follows the reference oracle /root/reference/_test.py:229-494
(test_rnea_grad_inner; Carpentier & Mansard, "Analytical Derivatives of Rigid Body Dynamics Algorithms").

Lane mapping: lane j of the solve's lane group owns the two derivative columns d/dq_j and d/dqd_j of every link
quantity (dv, da, df), as register 6-vectors.  A column is structurally zero at link i unless j is an ancestor of i,
i itself (forward sweep) or in i's subtree (after the backward sweep), and zeros propagate through the linear
recursions, so the reference's sparsity-compressed column bookkeeping (helpers/_topology_helpers.py:515-542,
_inverse_dynamics_gradient.py:597-651,731-760) and its shared-memory atomics (:653-655) are not needed.
The wave-uniform link quantities (v, a, f, I v) are recomputed in registers inside the same depth-first walk
(this is the RNEA-with-qdd pass a7 of SURVEY.md section 8(a)), and the forward and backward sweeps of the derivative
are fused into that one walk, so only O(depth) link vectors are live.

Identities used (checked against the oracle by the tests):
  * mxS(S_i, X_i v_parent) == mxS(S_i, v_i)    because v_i = X_i v_parent + S_i qd_i and S_i x S_i = 0
    (the oracle's MxXv, _test.py:308, therefore equals its Mxv, :310; both vanish for root joints);
  * (fx(v) I) dv == fx(v) (I dv)               (the oracle materialises FxvI = fx(v) I, _test.py:403).
"""


def gen_inverse_dynamics_gradient_inner_temp_mem_size(self):
    return 0


def gen_inverse_dynamics_gradient_kernel_max_temp_mem_size(self):
    return 0


def gen_inverse_dynamics_gradient_inner_function_call(self, use_thread_group=False, updated_var_names=None):
    self.gen_add_code_line("inverse_dynamics_gradient_inner<T>(dc_dq, dc_dqd, s_qd, s_qdd, s_X, s_F, gravity, lane);")


def gen_inverse_dynamics_gradient_inner(self, use_thread_group=False):
    m = self.model
    n = m.n
    self.gen_add_func_doc("Computes the gradient of inverse dynamics (this lane's two columns)",
                          ["dc_dq[i] = d c_i / d q_lane and dc_dqd[i] = d c_i / d qd_lane for i = 0..n-1 (zero for lanes >= n)",
                           "v, a, f of the RNEA pass with qdd are recomputed wave-uniformly inside the same tree walk",
                           "follows /root/reference/_test.py:229-494 incl. the damping term on diag(dc/dqd)"],
                          ["dc_dq, dc_dqd are the register outputs", "s_qd is the vector of joint velocities in LDS",
                           "s_qdd is the vector of joint accelerations in LDS", "s_X is this solve's compact X(q) storage",
                           "s_F is LDS scratch for the wave-uniform link forces (8 floats per joint)", "gravity is the gravity constant", "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void inverse_dynamics_gradient_inner(T (&dc_dq)[%d], T (&dc_dqd)[%d], const T *s_qd, const T *s_qdd, const T *s_X, T *s_F, const T gravity, const int lane) {" % (n, n), True)

    def pre(i):
        s, p = m.S_index[i], m.parent[i]
        I = str(i)
        P = str(p)
        self.gen_add_code_line("const T self_" + I + " = (lane == " + I + ") ? static_cast<T>(1) : static_cast<T>(0);")
        self.gen_add_code_line("T v_%s[6], a_%s[6], dvq_%s[6], dvd_%s[6], daq_%s[6], dad_%s[6], dfq_%s[6], dfd_%s[6];" % ((I,) * 8))
        self.gen_add_code_line("{", True)
        self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*" + I + "]);")
        self.gen_add_code_line("const T qd = s_qd[" + I + "]; const T qdd = s_qdd[" + I + "];")
        self.gen_add_code_line("// wave-uniform link quantities (RNEA with qdd)")
        self.gen_add_code_line("T Xa[6], Mxv[6], MxXa[6], Iv[6];")
        if p == -1:
            self.gen_add_code_line("grid_zero6(v_%s); v_%s[%d] = qd;" % (I, I, s))
            self.gen_add_code_line("grid_zero6(Xa); Xa[3] = X[2]*gravity; Xa[4] = X[5]*gravity; Xa[5] = X[8]*gravity;")
        else:
            self.gen_add_code_line("grid_xmul(v_%s, X, v_%s); v_%s[%d] += qd;" % (I, P, I, s))
            self.gen_add_code_line("grid_xmul(Xa, X, a_%s);" % P)
        self.gen_add_code_line("grid_zero6(Mxv); grid_mxS_peq<T,%d>(Mxv, v_%s, static_cast<T>(1));" % (s, I))
        self.gen_add_code_line("grid_zero6(MxXa); grid_mxS_peq<T,%d>(MxXa, Xa, static_cast<T>(1));" % s)
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { a_%s[r] = Xa[r] + Mxv[r]*qd; }" % I)
        self.gen_add_code_line("a_%s[%d] += qdd;" % (I, s))
        self.gen_add_code_line("// the link force is wave-uniform and only needed again on the way back up: park it in LDS instead of 6 VGPRs per level")
        self.gen_add_code_line("{ T f[6]; grid_imul_%s(Iv, v_%s); grid_imul_%s(f, a_%s); grid_fxv_peq(f, v_%s, Iv);" % (I, I, I, I, I))
        self.gen_add_code_line("  if (lane == 0) {")
        self.gen_add_code_line("      #pragma unroll")
        self.gen_add_code_line("      for (int r = 0; r < 6; r++) { s_F[%d + r] = f[r]; }" % (8 * i))
        self.gen_add_code_line("  } }")
        self.gen_add_code_line("// this lane's columns: dv, da (forward recursions with the self terms of column == joint)")
        if p == -1:
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { dvq_%s[r] = self_%s*Mxv[r]; dvd_%s[r] = static_cast<T>(0); daq_%s[r] = self_%s*MxXa[r]; dad_%s[r] = self_%s*Mxv[r]; }" % (I, I, I, I, I, I, I))
            self.gen_add_code_line("dvd_%s[%d] = self_%s;" % (I, s, I))
        else:
            self.gen_add_code_line("grid_xmul(dvq_%s, X, dvq_%s); grid_xmul(dvd_%s, X, dvd_%s);" % (I, P, I, P))
            self.gen_add_code_line("grid_xmul(daq_%s, X, daq_%s); grid_xmul(dad_%s, X, dad_%s);" % (I, P, I, P))
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { dvq_%s[r] += self_%s*Mxv[r]; daq_%s[r] += self_%s*MxXa[r]; dad_%s[r] += self_%s*Mxv[r]; }" % (I, I, I, I, I, I))
            self.gen_add_code_line("dvd_%s[%d] += self_%s;" % (I, s, I))
        self.gen_add_code_line("grid_mxS_peq<T,%d>(daq_%s, dvq_%s, qd); grid_mxS_peq<T,%d>(dad_%s, dvd_%s, qd);" % (s, I, I, s, I, I))
        self.gen_add_code_line("// df = I da + fx(dv) (I v) + fx(v) (I dv)")
        self.gen_add_code_line("{ T Idv[6]; grid_imul_%s(dfq_%s, daq_%s); grid_fxv_peq(dfq_%s, dvq_%s, Iv); grid_imul_%s(Idv, dvq_%s); grid_fxv_peq(dfq_%s, v_%s, Idv); }" % (I, I, I, I, I, I, I, I, I))
        self.gen_add_code_line("{ T Idv[6]; grid_imul_%s(dfd_%s, dad_%s); grid_fxv_peq(dfd_%s, dvd_%s, Iv); grid_imul_%s(Idv, dvd_%s); grid_fxv_peq(dfd_%s, v_%s, Idv); }" % (I, I, I, I, I, I, I, I, I))
        self.gen_add_end_control_flow()

    def post(i):
        s, p = m.S_index[i], m.parent[i]
        I = str(i)
        P = str(p)
        damp = m.damping[i]
        self.gen_add_code_line("dc_dq[%s] = dfq_%s[%d];" % (I, I, s))
        self.gen_add_code_line("dc_dqd[%s] = dfd_%s[%d]%s;" % (I, I, s, (" + self_" + I + "*static_cast<T>(" + repr(float(damp)) + ")") if damp != 0.0 else ""))
        if p != -1:
            self.gen_add_sync(use_thread_group)
            self.gen_add_code_line("{", True)
            self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*" + I + "]);")
            self.gen_add_code_line("T f[6], fp[6];")
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { f[r] = s_F[%d + r]; fp[r] = s_F[%d + r]; }" % (8 * i, 8 * p))
            self.gen_add_code_line("// column == joint picks up -X^T mxS(f) where f is the accumulated subtree force")
            self.gen_add_code_line("grid_mxS_peq<T,%d>(dfq_%s, f, -self_%s);" % (s, I, I))
            self.gen_add_code_line("grid_xtmul_peq(fp, X, f); grid_xtmul_peq(dfq_%s, X, dfq_%s); grid_xtmul_peq(dfd_%s, X, dfd_%s);" % (P, I, P, I))
            self.gen_add_code_line("if (lane == 0) {", True)
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { s_F[%d + r] = fp[r]; }" % (8 * p))
            self.gen_add_end_control_flow()
            self.gen_add_end_control_flow()

    self.gen_tree_traversal(pre, post)
    self.gen_add_end_function()


def gen_inverse_dynamics_gradient_kernel(self, use_thread_group=False, use_qdd_input=False, single_call_timing=False):
    n = self.model.n
    func_params = ["d_dc_du is a pointer to memory for the final result of size 2*NUM_JOINTS*NUM_JOINTS = " + str(2 * n * n),
                   "d_q_dq is the vector of joint positions and velocities", "stride_q_qd is the stride between each q, qd",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)"]
    func_def = "void inverse_dynamics_gradient_kernel(T *d_dc_du, const T *d_q_qd, const int stride_q_qd, "
    if use_qdd_input:
        func_def += "const T *d_qdd, "
        func_params.insert(-3, "d_qdd is the vector of joint accelerations")
    func_def += "const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {"
    if single_call_timing:
        func_def = func_def.replace("kernel(", "kernel_single_timing(")
    self.gen_add_func_doc("Computes the gradient of inverse dynamics", ["output layout d_dc_du[k*2n^2 + col*n + row], col in [0,2n) = [d/dq | d/dqd]"], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__global__ __launch_bounds__(GRID_MAX_THREADS)")
    self.gen_add_code_line(func_def, True)
    self.gen_kernel_prologue("GRID_LDS_PER_SOLVE")
    self.gen_add_code_lines(["T *s_q_qd = &s_mem[GRID_OFF_IN]; T *s_q = s_q_qd; T *s_qd = &s_q_qd[%d]; T *s_qdd = &s_q_qd[%d];" % (n, 2 * n),
                             "T *s_X = &s_mem[GRID_OFF_X]; T *s_F = &s_mem[GRID_OFF_F]; T *s_dc_du = &s_mem[GRID_OFF_OUT];"])
    if single_call_timing:
        self.gen_add_code_line("const int k = 0; const int kc = 0; const bool valid = (blockIdx.x + blockIdx.y == 0) && (grp == 0);")
        self.gen_add_code_line("if (!valid) {return;}")
    else:
        self.gen_add_parallel_loop("k", "NUM_TIMESTEPS", use_thread_group, block_level=True)
    if use_qdd_input:
        self.gen_kernel_load_inputs("q_qd", "stride_q_qd", 2 * n, use_thread_group, "qdd", n, n)
    else:
        self.gen_kernel_load_inputs("q_qd", "stride_q_qd", 2 * n, use_thread_group)
        self.gen_add_code_line("if (lane < %d) { s_qdd[lane] = static_cast<T>(0); }" % n)
    if single_call_timing:
        self.gen_add_code_line("for (int rep = 0; rep < NUM_TIMESTEPS; rep++){", True)
    self.gen_add_code_line("// compute")
    self.gen_load_update_XImats_helpers_function_call(use_thread_group)
    self.gen_add_code_line("T dc_dq[%d], dc_dqd[%d];" % (n, n))
    self.gen_inverse_dynamics_gradient_inner_function_call(use_thread_group)
    self.gen_add_code_line("if (lane < %d) {" % n, True)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int row = 0; row < %d; row++) { s_dc_du[lane*%d + row] = dc_dq[row]; s_dc_du[%d + lane*%d + row] = dc_dqd[row]; }" % (n, n, n * n, n))
    self.gen_add_end_control_flow()
    if single_call_timing:
        self.gen_add_end_control_flow()
    self.gen_kernel_save_result("dc_du", 2 * n * n, 2 * n * n, use_thread_group)
    if not single_call_timing:
        self.gen_add_end_control_flow()
    self.gen_add_end_function()


def gen_inverse_dynamics_gradient_host(self, mode=0):
    single_call_timing = mode == 1
    compute_only = mode == 2
    func_params = ["hd_data is the packaged input and output pointers",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant,",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)",
                   "streams are pointers to HIP streams for async memory transfers (if needed)"]
    name = "inverse_dynamics_gradient" + ("_single_timing" if single_call_timing else "") + ("_compute_only" if compute_only else "")
    self.gen_add_func_doc("Computes the gradient of inverse dynamics", [], func_params, None)
    self.gen_add_code_line("template <typename T, bool USE_QDD_FLAG = false, bool USE_COMPRESSED_MEM = false>")
    self.gen_add_code_line("__host__")
    self.gen_add_code_line("void " + name + "(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps,")
    self.gen_add_code_line("                      const dim3 block_dimms, const dim3 thread_dimms" + ("" if compute_only else ", hipStream_t *streams") + ") {", True)
    self.gen_add_code_line("int stride_q_qd = USE_COMPRESSED_MEM ? 2*NUM_JOINTS: 3*NUM_JOINTS;")
    cnt = "" if single_call_timing else "num_timesteps*"
    if not compute_only:
        self.gen_add_code_lines(["// start code with memory transfer",
                                 "if (USE_COMPRESSED_MEM) {gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd,hd_data->h_q_qd,stride_q_qd*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[0]));}",
                                 "else {gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q_qd*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[0]));}",
                                 "if (USE_QDD_FLAG) {gpuErrchk(hipMemcpyAsync(hd_data->d_qdd,hd_data->h_qdd,NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[1]));}",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    kern = "inverse_dynamics_gradient_kernel" + ("_single_timing" if single_call_timing else "") + "<T>"
    self.gen_add_code_line("// then call the kernel")
    self.gen_add_code_line("const T *d_in = USE_COMPRESSED_MEM ? hd_data->d_q_qd : hd_data->d_q_qd_u;")
    if single_call_timing:
        self.gen_add_code_line("struct timespec start, end; clock_gettime(CLOCK_MONOTONIC,&start);")
    self.gen_add_code_lines(["if (USE_QDD_FLAG) {hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,ID_DU_DYNAMIC_SHARED_MEM_COUNT*sizeof(T),0,hd_data->d_dc_du,d_in,stride_q_qd,hd_data->d_qdd,d_robotModel,gravity,num_timesteps);}",
                             "else {hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,ID_DU_DYNAMIC_SHARED_MEM_COUNT*sizeof(T),0,hd_data->d_dc_du,d_in,stride_q_qd,d_robotModel,gravity,num_timesteps);}",
                             "gpuErrchk(hipGetLastError()); gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("clock_gettime(CLOCK_MONOTONIC,&end);")
    if not compute_only:
        self.gen_add_code_lines(["// finally transfer the result back",
                                 "gpuErrchk(hipMemcpy(hd_data->h_dc_du,hd_data->d_dc_du,NUM_JOINTS*2*NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyDeviceToHost));",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("printf(\"Single Call ID_DU %fus\\n\",time_delta_us_timespec(start,end)/static_cast<double>(num_timesteps));")
    self.gen_add_end_function()


def gen_inverse_dynamics_gradient(self, use_thread_group=False):
    self.gen_inverse_dynamics_gradient_inner(use_thread_group)
    for use_qdd in (True, False):
        for timing in (True, False):
            self.gen_inverse_dynamics_gradient_kernel(use_thread_group, use_qdd, timing)
    for mode in (0, 1, 2):
        self.gen_inverse_dynamics_gradient_host(mode)
