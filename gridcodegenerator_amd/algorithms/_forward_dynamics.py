"""Forward dynamics emitter for the HIP/CDNA4 backend: qdd = M^-1 (u - c).

Mirrors the role of the reference's algorithms/_forward_dynamics.py (gen_forward_dynamics_finish :21-50,
gen_forward_dynamics_inner :73, device/kernel/host :110-255); mathematics as in /root/reference/_test.py:498-501.
"""


def gen_forward_dynamics_inner_temp_mem_size(self):
    return 0


def gen_forward_dynamics_finish_function_call(self, use_thread_group=False, updated_var_names=None):
    self.gen_add_code_line("forward_dynamics_finish<T>(s_qdd, s_u, c, s_Minv, lane);")
    self.gen_add_sync(use_thread_group)


def gen_forward_dynamics_finish(self, use_thread_group=False):
    n = self.model.n
    self.gen_add_func_doc("Finish the forward dynamics computation with qdd = Minv*(u-c)",
                          ["lane r computes qdd[r]; s_Minv is dense symmetric so no upper-triangle index trick is needed",
                           "the caller must grid_wave_sync() before s_qdd is read by other lanes"],
                          ["s_qdd is a pointer to LDS for the final result", "s_u is the vector of joint input torques in LDS",
                           "c is the bias force (registers, identical in every lane)", "s_Minv is the (dense symmetric) inverse mass matrix in LDS",
                           "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void forward_dynamics_finish(T *s_qdd, const T *s_u, const T (&c)[%d], const T *s_Minv, const int lane) {" % n, True)
    self.gen_add_code_line("if (lane < %d) {" % n, True)
    self.gen_add_code_line("T val = static_cast<T>(0);")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int col = 0; col < %d; col++) { val += s_Minv[lane*%d + col]*(s_u[col] - c[col]); }" % (n, self.minv_ld))
    self.gen_add_code_line("s_qdd[lane] = val;")
    self.gen_add_end_control_flow()
    self.gen_add_end_function()


def gen_forward_dynamics_inner_function_call(self, use_thread_group=False, updated_var_names=None):
    self.gen_add_code_line("forward_dynamics_inner<T>(s_qdd, s_qd, s_u, s_X, s_U, s_T, s_Minv, d_robotModel, gravity, lane);")


def gen_forward_dynamics_inner(self, use_thread_group=False):
    """Fused M^-1 + RNEA(qdd = 0) + finish.  The two algorithms are independent until qdd = M^-1 (u - c), and the M^-1 backward sweep
    is a chain of LDS round trips (U publish -> read, transpose write -> read) with little arithmetic in between, so the
    wave-uniform RNEA forward steps are emitted INSIDE those regions (one link per M^-1 joint) where they fill the latency
    bubbles; the RNEA backward steps share the region of the M^-1 forward sweep."""
    m = self.model
    n = m.n
    self.gen_add_func_doc("Computes forward dynamics", ["fused direct_minv_inner + inverse_dynamics_inner(qdd = 0) + forward_dynamics_finish; s_qdd and s_Minv are valid for all lanes on return"],
                          ["s_qdd is a pointer to LDS for the final result", "s_qd is the vector of joint velocities", "s_u is the vector of joint input torques",
                           "s_X is this solve's compact X(q) storage", "s_U, s_T are LDS scratch (see direct_minv_inner)", "s_Minv receives the dense inverse mass matrix",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU", "gravity is the gravity constant",
                           "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void forward_dynamics_inner(T *s_qdd, const T *s_qd, const T *s_u, const T *s_X, T *s_U, T *s_T, T *s_Minv, const robotModel<T> *d_robotModel, const T gravity, const int lane) {", True)
    self.gen_add_code_line("T c[%d];" % n)
    if (not self.tuning["fuse_fd"]) or n > 9:  # the fused form keeps every link's RNEA vectors live (18 VGPRs per joint): small robots only
        self.gen_direct_minv_inner_function_call(use_thread_group)
        self.gen_inverse_dynamics_inner_function_call(use_thread_group, True, False)
        self.gen_forward_dynamics_finish_function_call(use_thread_group)
        self.gen_add_end_function()
        return
    self.gen_add_code_line("T rv[%d][6], ra[%d][6], rf[%d][6]; // RNEA link vectors (wave-uniform)" % (n, n, n))
    order = []  # RNEA forward order = pre-order = joint id order

    def rnea_fwd_step(idx):
        j = idx  # ids are DFS pre-order: parents come first
        s, p = m.S_index[j], m.parent[j]
        self._cur_joint = j
        self.gen_add_code_line("{ // RNEA forward step of link %d, placed here to overlap the LDS round trips of the M^-1 sweep" % j, True)
        self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*%d]);" % j)
        self.gen_add_code_line("const T qd_r = s_qd[%d];" % j)
        if p == -1:
            self.gen_add_code_line("grid_zero6(rv[%d]); rv[%d][%d] = qd_r;" % (j, j, s))
            self.gen_add_code_line("grid_zero6(ra[%d]); ra[%d][3] = X[2]*gravity; ra[%d][4] = X[5]*gravity; ra[%d][5] = X[8]*gravity;" % (j, j, j, j))
        else:
            self.gen_add_code_line("grid_xmul(rv[%d], X, rv[%d]); rv[%d][%d] += qd_r;" % (j, p, j, s))
            self.gen_add_code_line("grid_xmul(ra[%d], X, ra[%d]); grid_mxS_peq<T,%d>(ra[%d], rv[%d], qd_r);" % (j, p, s, j, j))
        if not self.reuse_rnea:
            self.gen_add_code_line("T Iv[6]; grid_imul_%d(Iv, rv[%d]); grid_imul_%d(rf[%d], ra[%d]); grid_fxv_peq(rf[%d], rv[%d], Iv); grid_pin6(rf[%d]);" % (j, j, j, j, j, j, j, j))
        else:
            self.gen_add_code_line("T Iv[6], b[6]; grid_imul_%d(Iv, rv[%d]); grid_zero6(b); grid_fxv_peq(b, rv[%d], Iv); grid_imul_%d(rf[%d], ra[%d]);" % (j, j, j, j, j, j))
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { rf[%d][r] += b[r]; }" % j)
            self.gen_add_code_line("grid_pin6(rf[%d]);" % j)
            self.gen_add_code_line("// v, I v and fx(v) I v do not depend on qdd: keep them for the gradient walk (inverse_dynamics_gradient_inner_reuse)")
            self.gen_add_code_line("if (lane == 0) {", True)
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { s_U[%d + r] = rv[%d][r]; s_U[%d + r] = Iv[r]; s_U[%d + r] = b[r]; }" % (18 * j, j, 18 * j + 6, 18 * j + 12))
            self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()

    def rnea_bwd_all():
        self.gen_add_code_line("// RNEA backward sweep (children before parents): c = S^T f + damping*qd, f_parent += X^T f")
        for j in range(n - 1, -1, -1):
            s, p = m.S_index[j], m.parent[j]
            damp = m.damping[j]
            self._cur_joint = j
            self.gen_add_code_line("c[%d] = rf[%d][%d]%s; grid_pin(c[%d]);" % (j, j, s, (" + static_cast<T>(" + repr(float(damp)) + ")*s_qd[%d]" % j) if damp != 0.0 else "", j))
            if p != -1:
                self.gen_add_code_line("{ T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*%d]); grid_xtmul_peq(rf[%d], X, rf[%d]); grid_pin6(rf[%d]); }" % (j, p, j, p))
        self._cur_joint = None

    self.gen_direct_minv_inner(use_thread_group, body_only=True, bwd_hook=rnea_fwd_step, fwd_hook=rnea_bwd_all)
    self.gen_add_sync(use_thread_group)
    self.gen_forward_dynamics_finish_function_call(use_thread_group)
    self.gen_add_end_function()


def gen_forward_dynamics_device(self, use_thread_group=False):
    self.gen_add_func_doc("Computes forward dynamics: X(q) update + forward_dynamics_inner (lane-group cooperative)",
                          ["all lanes of the solve's lane group must call it; s_qdd is visible to the group on return"],
                          ["s_qdd is the output vector of joint accelerations in LDS", "s_q is the vector of joint positions", "s_qd is the vector of joint velocities",
                           "s_u is the vector of joint input torques", "s_work is this solve's LDS workspace of GRID_LDS_PER_SOLVE elements",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU", "gravity is the gravity constant",
                           "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void forward_dynamics_device(T *s_qdd, const T *s_q, const T *s_qd, const T *s_u, T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane, const int off_sp = GRID_OFF_SP) {", True)
    self.gen_add_code_line("T *s_X = &s_work[GRID_OFF_X]; T *s_U = &s_work[GRID_OFF_U]; T *s_T = &s_work[GRID_OFF_T]; T *s_Minv = &s_work[GRID_OFF_MINV];")
    self.gen_load_update_XImats_helpers_function_call(use_thread_group)
    if self.tip_frame:  # serial revolute chains: M from the tip-frame composites, factored in registers
        self.gen_add_code_line("(void)s_T;")
        self.gen_add_code_line("forward_dynamics_inner_tip<T>(s_qdd, s_qd, s_u, s_X, s_U, s_Minv, gravity, d_robotModel, lane);")
        self.gen_add_sync(use_thread_group)
        self.gen_add_end_function()
        return
    if getattr(self, "branch_components", False):  # branched revolute robots: M from the branch-frame composites, tree-sparse factorisation
        self.gen_add_code_line("(void)s_T; (void)s_U; (void)s_Minv;")
        self.gen_add_code_line("forward_dynamics_inner_branch<T>(s_qdd, s_qd, s_u, s_X, &s_work[off_sp], d_robotModel, gravity, lane);")
        self.gen_add_sync(use_thread_group)
        self.gen_add_end_function()
        return
    self.gen_forward_dynamics_inner_function_call(use_thread_group)
    self.gen_add_end_function()


def gen_forward_dynamics_kernel(self, use_thread_group=False, single_call_timing=False):
    n = self.model.n
    func_params = ["d_qdd is the vector of joint accelerations", "d_q_qd_u is the vector of joint positions, velocities, and input torques",
                   "stride_q_qd_u is the stride between each q, qd, u",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)"]
    func_def = "void forward_dynamics_kernel(T *d_qdd, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {"
    if single_call_timing:
        func_def = func_def.replace("kernel(", "kernel_single_timing(")
    self.gen_add_func_doc("Computes forward dynamics", [], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__global__ GRID_LAUNCH_BOUNDS")
    self.gen_add_code_line(func_def, True)
    self.gen_kernel_prologue("FD_LDS_PER_SOLVE")
    self.gen_add_code_lines(["T *s_q_qd_u = &s_mem[GRID_OFF_IN]; T *s_q = s_q_qd_u; T *s_qd = &s_q_qd_u[%d]; T *s_u = &s_q_qd_u[%d];" % (n, 2 * n),
                             "T *s_qdd = &s_out_all[grp*%d];" % n])
    if single_call_timing:
        self.gen_add_code_line("const int k = 0; const int kc = 0; const bool valid = (blockIdx.x + blockIdx.y == 0) && (grp == 0); const int lane = lane_id; const int NUM_TIMESTEPS_OUT = 1;")
        self.gen_add_code_line("if (!valid) {return;}")
    else:
        self.gen_add_parallel_loop("k", "NUM_TIMESTEPS", use_thread_group, block_level=True)
    self.gen_kernel_load_inputs("q_qd_u", "stride_q_qd_u", 3 * n, use_thread_group)
    if single_call_timing:
        self.gen_add_code_line("for (int rep = 0; rep < NUM_TIMESTEPS; rep++){", True)
    self.gen_add_code_line("// compute")
    self.gen_add_code_line("forward_dynamics_device<T>(s_qdd, s_q, s_qd, s_u, s_mem, d_robotModel, gravity, lane, FD_OFF_SP);")
    if single_call_timing:
        self.gen_add_end_control_flow()
    if single_call_timing:
        self.gen_kernel_save_result_single_timing("qdd", n, use_thread_group)
    else:
        self.gen_kernel_save_result("qdd", n, n, use_thread_group)
    if not single_call_timing:
        self.gen_add_end_control_flow()
    self.gen_add_end_function()


def gen_forward_dynamics_host(self, mode=0):
    single_call_timing = mode == 1
    compute_only = mode == 2
    func_params = ["hd_data is the packaged input and output pointers",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant,",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)",
                   "streams are pointers to HIP streams for async memory transfers (if needed)"]
    name = "forward_dynamics" + ("_single_timing" if single_call_timing else "") + ("_compute_only" if compute_only else "")
    self.gen_add_func_doc("Computes forward dynamics", [], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__host__")
    self.gen_add_code_line("void " + name + "(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps,")
    self.gen_add_code_line("                      const dim3 block_dimms, const dim3 thread_dimms" + ("" if compute_only else ", hipStream_t *streams") + ") {", True)
    self.gen_add_code_line("int stride_q_qd_u = 3*NUM_JOINTS;")
    cnt = "" if single_call_timing else "num_timesteps*"
    if not compute_only:
        self.gen_add_code_lines(["// start code with memory transfer",
                                 "gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q_qd_u*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[0]));",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    kern = "forward_dynamics_kernel" + ("_single_timing" if single_call_timing else "") + "<T>"
    self.gen_add_code_line("// then call the kernel")
    if single_call_timing:
        self.gen_add_code_line("struct timespec start, end; clock_gettime(CLOCK_MONOTONIC,&start);")
    self.gen_add_code_lines(["hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, FD_LDS_PER_SOLVE, FD_OUT_PER_SOLVE),0,hd_data->d_qdd,hd_data->d_q_qd_u,stride_q_qd_u,d_robotModel,gravity,num_timesteps);",
                             "gpuErrchk(hipGetLastError()); gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("clock_gettime(CLOCK_MONOTONIC,&end);")
    if not compute_only:
        self.gen_add_code_lines(["// finally transfer the result back",
                                 "gpuErrchk(hipMemcpy(hd_data->h_qdd,hd_data->d_qdd,NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyDeviceToHost));",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("printf(\"Single Call FD %fus\\n\",time_delta_us_timespec(start,end)/static_cast<double>(num_timesteps));")
    self.gen_add_end_function()


def gen_forward_dynamics(self, use_thread_group=False):
    self.gen_forward_dynamics_finish(use_thread_group)
    self.gen_forward_dynamics_inner(use_thread_group)
    self.gen_forward_dynamics_device(use_thread_group)
    self.gen_forward_dynamics_kernel(use_thread_group, True)
    self.gen_forward_dynamics_kernel(use_thread_group, False)
    for mode in (0, 1, 2):
        self.gen_forward_dynamics_host(mode)
