"""Articulated-Body Algorithm (O(n) forward dynamics), emitter for the HIP/CDNA4 backend.

Mirrors the role of the reference's algorithms/_aba.py (gen_aba_inner :1-420, device :446, kernel :482, host :539): qdd = FD(q, qd, tau).
The reference ships no NumPy oracle for ABA; the result is the same vector as the reference's forward dynamics
(/root/reference/_test.py:498-501, qdd = Minv (tau - c)), which the goldens pin - that is what the tests compare against.
Differences from the reference's emitted CUDA: any revolute/prismatic joint axis (the reference
hard-codes revolute-z through mx2_scaled, _aba.py:123), output stride n (the reference passes "1", :524), velocity damping as in the
oracle's c.

Featherstone's three passes in the lane-group model (lane j <-> joint j, lanes 0..5 also own one column of every articulated inertia):
  pass 1+2 are ONE depth-first walk.  Going down: v_i = X v_p + S qd_i, pA_i = v x* I v, IA_i = I_i.  Coming back up:
      U = IA S (broadcast from the lane that owns that column), D = S.U, u = tau - damping*qd - S.pA,
      Ia = IA - U U^T / D, pa = pA + Ia c + U u / D with c = v x S qd,  IA_p += X^T Ia X,  pA_p += X^T pa.
      v, pA, c and u are wave-uniform (every lane computes them); Ia c needs all six columns: lane r dots its column with c and the six
      results are gathered with lane shuffles; X^T Ia X is the two one-sided products with a 6x6 transpose through LDS of direct_minv_inner.
  pass 3 is a second walk: a' = X a_p + c, qdd_i = (u - U.a') / D, a_i = a' + S qdd_i  (v is recomputed: 14 FMAs per joint instead of 6
      live registers per tree level).  U_i, 1/D_i, u_i wait in registers (n <= 9) or in the U scratch of the LDS slice.
"""


def gen_aba_inner_temp_mem_size(self):
    return 0  # LDS needs are part of the fixed per-solve slice


def gen_aba_inner_function_call(self, use_thread_group=False, updated_var_names=None):
    self.gen_add_code_line("aba_inner<T>(s_qdd, static_cast<T *>(nullptr), s_qd, s_tau, s_X, s_U, s_T, d_robotModel, gravity, lane);")
    self.gen_add_sync(use_thread_group)


def gen_aba_inner(self, use_thread_group=False):
    m = self.model
    n = m.n
    G = self.lanes_per_solve
    IA0 = 0 if self.cols_per_lane == 2 else G // 2
    keep = n <= 9
    self.gen_add_func_doc("Compute the ABA (Articulated Body Algorithm)",
                          ["qdd = FD(q, qd, tau) in O(n); same result as forward_dynamics_inner (the oracle's qdd = Minv (tau - c), /root/reference/_test.py:498-501)",
                           "lane 0 writes s_qdd (and s_va when it is not nullptr); the caller must grid_wave_sync() before other lanes read them"],
                          ["s_qdd is the output vector of joint accelerations in LDS",
                           "s_va receives the link velocities and accelerations [v_0..v_{n-1} | a_0..a_{n-1}] (6 values each), or nullptr",
                           "s_qd is the vector of joint velocities in LDS", "s_tau is the vector of joint torques in LDS",
                           "s_X is this solve's compact X(q) storage", "s_U is LDS scratch: per joint U (6), 1/D, u", "s_T is LDS scratch for the 6x6 transpose",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU", "gravity is the gravity constant",
                           "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void aba_inner(T *s_qdd, T *s_va, const T *s_qd, const T *s_tau, const T *s_X, T *s_U, T *s_T, const robotModel<T> *d_robotModel, const T gravity, const int lane) {", True)
    if IA0 == 0:
        self.gen_add_code_line("const int cI = lane < 5 ? lane : 5; // articulated-inertia column owned by this lane (lanes 0..5 are live)")
        self.gen_add_code_line("const bool isIA = lane < 6;")
    else:
        self.gen_add_code_line("const int cI = (lane < %d) ? 0 : ((lane > %d) ? 5 : (lane - %d));" % (IA0, IA0 + 5, IA0))
        self.gen_add_code_line("const bool isIA = (lane >= %d) && (lane < %d);" % (IA0, IA0 + 6))
    self.gen_add_code_line("const T *d_I = &grid_model_constants(static_cast<const T *>(nullptr))[" + str(18 * n) + " + 6*cI]; (void)d_robotModel;")
    if keep:
        self.gen_add_code_line("T Uk[%d][6], Dk[%d], uk[%d]; (void)s_U; // U_i, 1/D_i, u_i of every joint, kept for the last pass" % (n, n, n))
    self.gen_add_code_line("//")
    self.gen_add_code_line("// passes 1 and 2: one depth-first walk (down: v, pA, IA = I; up: U, D, u, articulated inertia and bias force to the parent)")
    self.gen_add_code_line("//")

    def pre(i):
        s, p = m.S_index[i], m.parent[i]
        self.gen_add_code_line("const T qd_%d = s_qd[%d];" % (i, i))
        self.gen_add_code_line("T v_%d[6], pA_%d[6], IA_%d[6];" % (i, i, i))
        self.gen_add_code_line("{", True)
        if p == -1:
            self.gen_add_code_line("grid_zero6(v_%d); v_%d[%d] = qd_%d;" % (i, i, s, i))
        else:
            self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*%d]);" % i)
            self.gen_add_code_line("grid_xmul(v_%d, X, v_%d); v_%d[%d] += qd_%d;" % (i, p, i, s, i))
        self.gen_add_code_line("T Iv[6]; grid_imul_%d(Iv, v_%d); grid_zero6(pA_%d); grid_fxv_peq(pA_%d, v_%d, Iv); grid_pin6(pA_%d);" % (i, i, i, i, i, i))
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { IA_%d[r] = d_I[%d + r]; }" % (i, 36 * i))
        self.gen_add_end_control_flow()

    def post(i):
        s, p = m.S_index[i], m.parent[i]
        damp = m.damping[i]
        tbuf = (m.depth[i] & 1) * 40
        self.gen_add_code_line("{", True)
        self.gen_add_code_line("T U[6];")
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { U[r] = grid_group_shfl(IA_%d[r], %d); }" % (i, IA0 + s))
        self.gen_add_code_line("const T Dinv = grid_rcp(U[%d]);" % s)
        self.gen_add_code_line("const T u = s_tau[%d]%s - pA_%d[%d];" % (i, (" - static_cast<T>(" + repr(float(damp)) + ")*qd_%d" % i) if damp != 0.0 else "", i, s))
        if keep:
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { Uk[%d][r] = U[r]; }" % i)
            self.gen_add_code_line("Dk[%d] = Dinv; uk[%d] = u;" % (i, i))
        else:
            self.gen_add_code_line("if (lane == 0) {", True)
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { s_U[%d + r] = U[r]; }" % (8 * i))
            self.gen_add_code_line("s_U[%d] = Dinv; s_U[%d] = u;" % (8 * i + 6, 8 * i + 7))
            self.gen_add_end_control_flow()
        if p != -1:
            self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*%d]);" % i)
            self.gen_add_code_line("T c[6]; grid_zero6(c); grid_mxS_peq<T,%d>(c, v_%d, qd_%d); // c = v x S qd" % (s, i, i))
            self.gen_add_code_line("T Ia[6], pa[6], Tc[6], Tr[6];")
            self.gen_add_code_line("const T w = Dinv*IA_%d[%d];" % (i, s))
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { Ia[r] = IA_%d[r] - U[r]*w; }" % i)
            self.gen_add_code_line("const T yl = grid_dot6(Ia, c); // (Ia c)[cI]: Ia is symmetric, so its row cI is this lane's column")
            self.gen_add_code_line("const T ud = u*Dinv;")
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { pa[r] = pA_%d[r] + grid_group_shfl(yl, %d + r) + U[r]*ud; }" % (i, IA0))
            self.gen_add_code_line("grid_xtmul_peq(pA_%d, X, pa); grid_pin6(pA_%d);" % (p, p))
            self.gen_add_code_line("// IA_parent += X^T Ia X, one column per lane, transposed through LDS")
            self.gen_add_code_line("grid_xtmul(Tc, X, Ia);")
            self.gen_add_code_line("if (isIA) {", True)
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { s_T[%d + 6*r + cI] = Tc[r]; }" % tbuf)
            self.gen_add_end_control_flow()
            self.gen_add_sync(use_thread_group)
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { Tr[r] = s_T[%d + 6*cI + r]; }" % tbuf)
            self.gen_add_code_line("grid_xtmul_peq(IA_%d, X, Tr); grid_pin6(IA_%d);" % (p, p))
        self.gen_add_end_control_flow()

    self.gen_tree_traversal(pre, post)
    self.gen_add_sync(use_thread_group)
    self.gen_add_code_line("//")
    self.gen_add_code_line("// pass 3: a' = X a_parent + c, qdd = (u - U.a')/D, a = a' + S qdd")
    self.gen_add_code_line("//")

    def pre3(i):
        s, p = m.S_index[i], m.parent[i]
        self.gen_add_code_line("T w_%d[6], a_%d[6];" % (i, i))
        self.gen_add_code_line("{", True)
        self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*%d]);" % i)
        self.gen_add_code_line("const T qd = s_qd[%d];" % i)
        if p == -1:
            self.gen_add_code_line("grid_zero6(w_%d); w_%d[%d] = qd;" % (i, i, s))
            self.gen_add_code_line("grid_zero6(a_%d); a_%d[3] = X[2]*gravity; a_%d[4] = X[5]*gravity; a_%d[5] = X[8]*gravity; // X*[0,0,0,0,0,g]" % (i, i, i, i))
        else:
            self.gen_add_code_line("grid_xmul(w_%d, X, w_%d); w_%d[%d] += qd;" % (i, p, i, s))
            self.gen_add_code_line("grid_xmul(a_%d, X, a_%d);" % (i, p))
        self.gen_add_code_line("grid_mxS_peq<T,%d>(a_%d, w_%d, qd);" % (s, i, i))
        if keep:
            self.gen_add_code_line("const T qdd = (uk[%d] - grid_dot6(Uk[%d], a_%d))*Dk[%d];" % (i, i, i, i))
        else:
            self.gen_add_code_line("T U[6];")
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { U[r] = s_U[%d + r]; }" % (8 * i))
            self.gen_add_code_line("const T qdd = (s_U[%d] - grid_dot6(U, a_%d))*s_U[%d];" % (8 * i + 7, i, 8 * i + 6))
        self.gen_add_code_line("a_%d[%d] += qdd; grid_pin6(a_%d);" % (i, s, i))
        self.gen_add_code_line("if (lane == 0) {", True)
        self.gen_add_code_line("s_qdd[%d] = qdd;" % i)
        self.gen_add_code_line("if (s_va != nullptr) {", True)
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { s_va[%d + r] = w_%d[r]; s_va[%d + r] = a_%d[r]; }" % (6 * i, i, 6 * n + 6 * i, i))
        self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()

    def post3(i):
        pass

    self.gen_tree_traversal(pre3, post3)
    self.gen_add_end_function()


def gen_aba_device(self, use_thread_group=False):
    self.gen_add_func_doc("Compute the ABA (Articulated Body Algorithm): X(q) update + aba_inner (lane-group cooperative)",
                          ["all lanes of the solve's lane group must call it; s_qdd is visible to the group on return"],
                          ["s_qdd is the output vector of joint accelerations in LDS", "s_q is the vector of joint positions", "s_qd is the vector of joint velocities",
                           "s_tau is the vector of joint torques", "s_work is this solve's LDS workspace of GRID_LDS_PER_SOLVE elements",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU", "gravity is the gravity constant",
                           "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void aba_device(T *s_qdd, const T *s_q, const T *s_qd, const T *s_tau, T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane) {", True)
    self.gen_add_code_line("T *s_X = &s_work[GRID_OFF_X]; T *s_U = &s_work[GRID_OFF_U]; T *s_T = &s_work[GRID_OFF_T];")
    self.gen_load_update_XImats_helpers_function_call(use_thread_group)
    self.gen_aba_inner_function_call(use_thread_group)
    self.gen_add_end_function()


def gen_aba_kernel(self, use_thread_group=False, single_call_timing=False):
    n = self.model.n
    func_params = ["d_qdd is the vector of joint accelerations", "d_q_qd_tau is the vector of joint positions, velocities, and torques",
                   "stride_q_qd is the stride between each q, qd, tau",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)"]
    func_def = "void aba_kernel(T *d_qdd, const T *d_q_qd_tau, const int stride_q_qd, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {"
    if single_call_timing:
        func_def = func_def.replace("kernel(", "kernel_single_timing(")
    self.gen_add_func_doc("Compute the ABA (Articulated Body Algorithm)", ["output layout d_qdd[k*n + i] (the reference's kernel passes stride \"1\" to its save helper, _aba.py:524)"], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__global__ GRID_LAUNCH_BOUNDS")
    self.gen_add_code_line(func_def, True)
    self.gen_kernel_prologue("ABA_LDS_PER_SOLVE")
    self.gen_add_code_lines(["T *s_q_qd_tau = &s_mem[GRID_OFF_IN]; T *s_q = s_q_qd_tau; T *s_qd = &s_q_qd_tau[%d]; T *s_tau = &s_q_qd_tau[%d];" % (n, 2 * n),
                             "T *s_qdd = &s_out_all[grp*%d];" % n])
    if single_call_timing:
        self.gen_add_code_line("const int k = 0; const int kc = 0; const bool valid = (blockIdx.x + blockIdx.y == 0) && (grp == 0); const int lane = lane_id; const int NUM_TIMESTEPS_OUT = 1;")
        self.gen_add_code_line("if (!valid) {return;}")
    else:
        self.gen_add_parallel_loop("k", "NUM_TIMESTEPS", use_thread_group, block_level=True)
    self.gen_kernel_load_inputs("q_qd_tau", "stride_q_qd", 3 * n, use_thread_group)
    if single_call_timing:
        self.gen_add_code_line("for (int rep = 0; rep < NUM_TIMESTEPS; rep++){", True)
    self.gen_add_code_line("// compute")
    self.gen_add_code_line("aba_device<T>(s_qdd, s_q, s_qd, s_tau, s_mem, d_robotModel, gravity, lane);")
    if single_call_timing:
        self.gen_add_end_control_flow()
    if single_call_timing:
        self.gen_kernel_save_result_single_timing("qdd", n, use_thread_group)
    else:
        self.gen_kernel_save_result("qdd", n, n, use_thread_group)
    if not single_call_timing:
        self.gen_add_end_control_flow()
    self.gen_add_end_function()


def gen_aba_host(self, mode=0):
    single_call_timing = mode == 1
    compute_only = mode == 2
    func_params = ["hd_data is the packaged input and output pointers",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant,",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)",
                   "streams are pointers to HIP streams for async memory transfers (if needed)"]
    name = "aba" + ("_single_timing" if single_call_timing else "") + ("_compute_only" if compute_only else "")
    self.gen_add_func_doc("Compute the ABA (Articulated Body Algorithm)", [], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__host__")
    self.gen_add_code_line("void " + name + "(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps,")
    self.gen_add_code_line("                      const dim3 block_dimms, const dim3 thread_dimms" + ("" if compute_only else ", hipStream_t *streams") + ") {", True)
    self.gen_add_code_line("int stride_q_qd = 3*NUM_JOINTS;")
    cnt = "" if single_call_timing else "num_timesteps*"
    if not compute_only:
        self.gen_add_code_lines(["// start code with memory transfer",
                                 "gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q_qd*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[0]));",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    kern = "aba_kernel" + ("_single_timing" if single_call_timing else "") + "<T>"
    self.gen_add_code_line("// then call the kernel")
    if single_call_timing:
        self.gen_add_code_line("struct timespec start, end; clock_gettime(CLOCK_MONOTONIC,&start);")
    self.gen_add_code_lines(["hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, ABA_LDS_PER_SOLVE, ABA_OUT_PER_SOLVE),0,hd_data->d_qdd,hd_data->d_q_qd_u,stride_q_qd,d_robotModel,gravity,num_timesteps);",
                             "gpuErrchk(hipGetLastError()); gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("clock_gettime(CLOCK_MONOTONIC,&end);")
    if not compute_only:
        self.gen_add_code_lines(["// finally transfer the result back",
                                 "gpuErrchk(hipMemcpy(hd_data->h_qdd,hd_data->d_qdd,NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyDeviceToHost));",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("printf(\"Single Call ABA %fus\\n\",time_delta_us_timespec(start,end)/static_cast<double>(num_timesteps));")
    self.gen_add_end_function()


def gen_aba(self, use_thread_group=False):
    self.gen_aba_inner(use_thread_group)
    self.gen_aba_device(use_thread_group)
    self.gen_aba_kernel(use_thread_group, True)
    self.gen_aba_kernel(use_thread_group, False)
    for mode in (0, 1, 2):
        self.gen_aba_host(mode)
