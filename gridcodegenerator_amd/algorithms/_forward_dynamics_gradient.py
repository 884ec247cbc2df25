"""Forward dynamics gradient emitter for the HIP/CDNA4 backend - THE hot path.

df/du = -M^-1 * dc/du  with u = (q, qd), qdd = M^-1 (tau - c).

Mirrors the role of the reference's algorithms/_forward_dynamics_gradient.py (inner call sequence :7-62, device
:64-105, kernel :113-184, host :186-249, driver :256-268) and follows the reference oracle
/root/reference/_test.py:496-520 (test_fd_grad).

Per solve (one lane group, lane j <-> joint j):
  load (q,qd,u) -> X(q) update -> direct_minv_inner -> inverse_dynamics_inner (qdd = 0) -> forward_dynamics_finish
  -> inverse_dynamics_gradient_inner (RNEA with qdd fused in) -> this lane's two columns of -M^-1 dc/du -> LDS staging
  -> coalesced store of the 2 n^2 record.

Intentional deviation from the reference's emitted CUDA: the reference's final loop also
writes M^-1 into s_temp[ind + n^2] from every thread, which races with the d/dqd half of the result; we output
-M^-1 dc/dq | -M^-1 dc/dqd (what the oracle returns, _test.py:504) and nothing else.
"""


def gen_forward_dynamics_gradient_inner_temp_mem_size(self, use_qdd_Minv_input=False):
    return 0


def gen_forward_dynamics_gradient_kernel_max_temp_mem_size(self):
    return 0


def gen_forward_dynamics_gradient_inner_python(self, use_thread_group=False, use_qdd_Minv_input=False, s_df_du_name="s_df_du"):
    """Emits the body shared by the device functions and the kernels; expects s_q/s_qd/(s_u)/s_qdd/s_Minv/s_X/s_U/s_T to be bound."""
    n = self.model.n
    stop = int(self.tuning["debug_stop"])  # timing ablation only (results are wrong when set): 1 = X update, 2 = +Minv, 3 = +RNEA/qdd, 4 = +gradient walk
    if getattr(self, "branch_frame", False) and stop in (0, 20) and not use_qdd_Minv_input:  # trees of revolute joints: one fused inner, every branch in its tip link's frame
        self.gen_forward_dynamics_gradient_inner_branch_function_call(use_thread_group, s_df_du_name)
        return
    if getattr(self, "branch_components", False) and stop == 0 and use_qdd_Minv_input:
        # (qdd, Minv) given: dc/du on the branch-frame path, then this lane's two columns times the caller's M^-1 (dense, read wave-uniformly from LDS)
        ld = self.minv_ld
        self.gen_add_code_line("inverse_dynamics_gradient_inner_branch<T>(%s, s_qd, s_qdd, s_X, &s_work[GRID_OFF_SP], d_robotModel, gravity, lane);" % s_df_du_name)
        self.gen_add_sync(use_thread_group)
        self.gen_add_code_line("if (lane < %d) {" % n, True)
        self.gen_add_code_line("T cq[%d], cd[%d];" % (n, n))
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int i = 0; i < %d; i++) { cq[i] = %s[lane*%d + i]; cd[i] = %s[(%d + lane)*%d + i]; }" % (n, s_df_du_name, n, s_df_du_name, n, n))
        self.gen_add_code_line("#pragma unroll 2")
        self.gen_add_code_line("for (int row = 0; row < %d; row++) {" % n, True)
        self.gen_add_code_line("T vq = static_cast<T>(0), vd = static_cast<T>(0);")
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int i = 0; i < %d; i++) { const T mi = s_Minv[row*%d + i]; vq += mi*cq[i]; vd += mi*cd[i]; }" % (n, ld))
        self.gen_add_code_line("%s[lane*%d + row] = -vq; %s[(%d + lane)*%d + row] = -vd;" % (s_df_du_name, n, s_df_du_name, n, n))
        self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
        return
    if self.tip_frame and stop in (0, 5, 6, 7, 20, 21):  # serial revolute chains: everything after the X update is one fused inner in the tip link's frame
        self.gen_forward_dynamics_gradient_inner_tip_function_call(use_thread_group, use_qdd_Minv_input, s_df_du_name)
        return
    if stop == 1:
        self.gen_add_code_line("if (lane < %d) { %s[lane] = s_X[GRID_X_STRIDE*lane]; }" % (n, s_df_du_name))
        return
    if stop == 2:
        self.gen_direct_minv_inner_function_call(use_thread_group)
        self.gen_add_code_line("if (lane < %d) { %s[lane] = s_Minv[lane]; }" % (n, s_df_du_name))
        return
    if not use_qdd_Minv_input:
        self.gen_forward_dynamics_inner_function_call(use_thread_group)  # Minv, c, qdd (v is recomputed below inside the fused gradient walk)
    if stop == 3:
        self.gen_add_code_line("if (lane < %d) { %s[lane] = s_qdd[lane]; }" % (n, s_df_du_name))
        return
    self.gen_add_code_line(self.gen_gradient_outputs_decl())
    self.gen_inverse_dynamics_gradient_inner_function_call(use_thread_group, reuse=not use_qdd_Minv_input)
    if stop == 4:
        self.gen_dc_du_to_lds(s_df_du_name)
        return
    self.gen_add_code_line("// finally df/du = -Minv*dc/du for the column(s) this lane owns (Minv is read wave-uniformly from LDS)")
    self.gen_dc_du_to_lds(s_df_du_name, minv_name="s_Minv")


def gen_forward_dynamics_gradient_device(self, use_thread_group=False, use_qdd_Minv_input=False):
    n = self.model.n
    func_params = ["s_df_du is a pointer to LDS for the final result of size 2*NUM_JOINTS*NUM_JOINTS = " + str(2 * n * n),
                   "s_q is the vector of joint positions", "s_qd is the vector of joint velocities",
                   "s_work is this solve's LDS workspace of GRID_LDS_PER_SOLVE elements (may contain s_q/s_qd/... in its GRID_OFF_IN part)",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant", "lane is the caller's lane index inside the solve's lane group"]
    func_def = "void forward_dynamics_gradient_device(T *s_df_du, const T *s_q, const T *s_qd, "
    if use_qdd_Minv_input:
        func_def += "const T *s_qdd, const T *s_Minv, "
        func_params.insert(3, "s_qdd is the vector of joint accelerations")
        func_params.insert(4, "s_Minv is the dense symmetric inverse mass matrix")
    else:
        func_def += "const T *s_u, "
        func_params.insert(3, "s_u is the vector of input torques")
    fused_out = self.tip_frame and not getattr(self, "branch_frame", False) and not use_qdd_Minv_input  # the tip-frame inner can leave qdd and M^-1 behind (what fdsva_so needs besides the gradient)
    func_def += "T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane" + (", T *s_qdd_out = nullptr, T *s_Minv_out = nullptr" if fused_out else "") + ") {"
    if fused_out:
        func_params.append("s_qdd_out, s_Minv_out (optional): where to leave qdd = FD(q, qd, u) and the dense M^-1 (leading dimension GRID_MINV_LD; may be &s_work[GRID_OFF_MINV])")
    self.gen_add_func_doc("Computes the gradient of forward dynamics (lane-group cooperative, for use inside other kernels)",
                          ["Uses the fd/du = -Minv*id/du trick as described in Carpentier and Mansard 'Analytical Derivatives of Rigid Body Dynamics Algorithms'",
                           "all lanes of the solve's lane group must call it; results are visible to the group after grid_wave_sync()"], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line(func_def, True)
    branch = getattr(self, "branch_frame", False) and not use_qdd_Minv_input
    if branch:
        self.gen_add_code_line("// (this path only touches the first FD_DU_LDS_PER_SOLVE elements of s_work)")
        self.gen_add_code_line("T *s_X = &s_work[GRID_OFF_X]; T *s_SP = &s_work[FD_DU_OFF_SP]; T *s_qdd = &s_work[FD_DU_OFF_QDD];")
    else:
        self.gen_add_code_line("T *s_X = &s_work[GRID_OFF_X]; T *s_U = &s_work[GRID_OFF_U]; T *s_T = &s_work[GRID_OFF_T]; T *s_F = &s_work[GRID_OFF_F]; T *s_J = &s_work[GRID_OFF_J];")
        if not use_qdd_Minv_input:
            self.gen_add_code_line("T *s_Minv = &s_work[GRID_OFF_MINV]; T *s_qdd = &s_work[GRID_OFF_QDD];")
        self.gen_add_code_line("(void)s_U; (void)s_T;")
    self.gen_load_update_XImats_helpers_function_call(use_thread_group)
    self.gen_forward_dynamics_gradient_inner_python(use_thread_group, use_qdd_Minv_input)
    self.gen_add_sync(use_thread_group)
    self.gen_add_end_function()


def gen_forward_dynamics_gradient_stream_device(self, use_thread_group=False):
    """Device function behind the forward_dynamics_gradient kernel where gen_lds_layout chose the streamed output (self.fd_stream_out)."""
    n = self.model.n
    self.gen_add_func_doc("Computes the gradient of forward dynamics and stores it to global memory half by half (lane-group cooperative; the kernel's form where LDS capacity bounds the resident waves)",
                          ["one half of the record is staged at a time (the dc/dqd columns wait in s_work[FD_DU_OFF_YPARK..] while dc/dq is assembled); all lanes of the solve's lane group must call it"],
                          ["d_df_du_k is this solve's record in global memory, 2*NUM_JOINTS*NUM_JOINTS values [col*n + row] (nullptr: lane group without a solve, nothing is stored)",
                           "s_half is LDS for one half of the record: NUM_JOINTS*NUM_JOINTS = " + str(n * n) + " values",
                           "s_q is the vector of joint positions", "s_qd is the vector of joint velocities", "s_u is the vector of input torques",
                           "s_work is this solve's LDS workspace of FD_DU_LDS_PER_SOLVE elements",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                           "gravity is the gravity constant", "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void forward_dynamics_gradient_stream_device(T *d_df_du_k, T *s_half, const T *s_q, const T *s_qd, const T *s_u, T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane) {", True)
    self.gen_add_code_line("T *s_X = &s_work[GRID_OFF_X]; T *s_SP = &s_work[FD_DU_OFF_SP]; T *s_qdd = &s_work[FD_DU_OFF_QDD];")
    self.gen_load_update_XImats_helpers_function_call(use_thread_group)
    self.gen_add_code_line("forward_dynamics_gradient_inner_branch_stream<T>(d_df_du_k, s_half, &s_work[FD_DU_OFF_YPARK], s_qd, s_u, s_X, s_SP, s_qdd, d_robotModel, gravity, lane);")
    self.gen_add_sync(use_thread_group)
    self.gen_add_end_function()


def gen_forward_dynamics_gradient_kernel(self, use_thread_group=False, use_qdd_Minv_input=False, single_call_timing=False):
    n = self.model.n
    stream = getattr(self, "fd_stream_out", False) and not use_qdd_Minv_input and not single_call_timing
    func_params = ["d_df_du is a pointer to memory for the final result of size 2*NUM_JOINTS*NUM_JOINTS = " + str(2 * n * n),
                   "d_q_dq is the vector of joint positions and velocities", "stride_q_qd is the stride between each q, qd",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)"]
    func_def = "void forward_dynamics_gradient_kernel(T *d_df_du, const T *d_q_qd, const int stride_q_qd, "
    if use_qdd_Minv_input:
        func_def += "const T *d_qdd, const T *d_Minv, "
        func_params.insert(-3, "d_qdd is the vector of joint accelerations")
        func_params.insert(-3, "d_Minv is the inverse mass matrix (column-major, only the upper triangle is read)")
    else:
        func_def = func_def.replace("_q_qd", "_q_qd_u")
        func_params[1] = "d_q_qd_u is the vector of joint positions, velocities, and input torques"
        func_params[2] = "stride_q_qd_u is the stride between each q, qd, u"
    func_def += "const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {"
    if single_call_timing:
        func_def = func_def.replace("kernel(", "kernel_single_timing(")
    self.gen_add_func_doc("Computes the gradient of forward dynamics",
                          ["output layout d_df_du[k*2n^2 + col*n + row], col in [0,2n) = [d qdd/dq | d qdd/dqd] (column-major n x 2n)"], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__global__ GRID_LAUNCH_BOUNDS")
    self.gen_add_code_line(func_def, True)
    self.gen_kernel_prologue("GRID_LDS_PER_SOLVE" if use_qdd_Minv_input else "FD_DU_LDS_PER_SOLVE")
    if int(self.tuning["stagger"]) > 0 and not use_qdd_Minv_input and not single_call_timing:
        self.gen_add_code_line("#if defined(__HIP_DEVICE_COMPILE__)")
        self.gen_add_code_line("if (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1) { __builtin_amdgcn_s_sleep(%d); } // HW_ID.wave_id: the second wave of a SIMD starts late" % min(127, int(self.tuning["stagger"])))
        self.gen_add_code_line("#endif")
    if self.tuning["debug_stop"] == 9:  # timing ablation only: an empty kernel (launch + dispatch cost)
        self.gen_add_code_line("if (NUM_TIMESTEPS > -1) {return;}")
    if use_qdd_Minv_input:
        self.gen_add_code_line("T *s_q_qd = &s_mem[GRID_OFF_IN]; T *s_q = s_q_qd; T *s_qd = &s_q_qd[%d]; T *s_qdd = &s_mem[GRID_OFF_QDD];" % n)
    else:
        self.gen_add_code_line("T *s_q_qd_u = &s_mem[GRID_OFF_IN]; T *s_q = s_q_qd_u; T *s_qd = &s_q_qd_u[%d]; T *s_u = &s_q_qd_u[%d]; T *s_qdd = &s_mem[%s];" % (n, 2 * n, "FD_DU_OFF_QDD" if getattr(self, "branch_frame", False) else "GRID_OFF_QDD"))
    if getattr(self, "branch_frame", False) and not use_qdd_Minv_input:
        self.gen_add_code_line("T *s_df_du = &s_out_all[grp*%d]; (void)s_qdd;%s" % (n * n if (self.tuning["out_half"] or stream) else 2 * n * n, " // ONE half of this solve's record at a time" if stream else ""))
    else:
        self.gen_add_code_line("T *s_Minv = &s_mem[GRID_OFF_MINV]; T *s_df_du = &s_out_all[grp*%d]; (void)s_Minv; (void)s_qdd;" % (2 * n * n))
    if single_call_timing:
        self.gen_add_code_line("const int k = 0; const int kc = 0; const bool valid = (blockIdx.x + blockIdx.y == 0) && (grp == 0); const int lane = lane_id; const int NUM_TIMESTEPS_OUT = 1;")
        self.gen_add_code_line("if (!valid) {return;}")
    else:
        self.gen_add_parallel_loop("k", "NUM_TIMESTEPS", use_thread_group, block_level=True)
    if use_qdd_Minv_input:
        self.gen_kernel_load_inputs("q_qd", "stride_q_qd", 2 * n, use_thread_group, "qdd", n, n, "Minv", n * n, n * n, symmetrize3=n)
    else:
        self.gen_kernel_load_inputs("q_qd_u", "stride_q_qd_u", 3 * n, use_thread_group)
    if single_call_timing:
        self.gen_add_code_line("// compute with NUM_TIMESTEPS as NUM_REPS for timing")
        self.gen_add_code_line("for (int rep = 0; rep < NUM_TIMESTEPS; rep++){", True)
    self.gen_add_code_line("// compute")
    if stream:
        self.gen_add_code_line("forward_dynamics_gradient_stream_device<T>(valid ? &d_df_du[static_cast<size_t>(kc)*%d] : nullptr, s_df_du, s_q, s_qd, s_u, s_mem, d_robotModel, gravity, lane); // (stores both halves itself)" % (2 * n * n))
    else:
        self.gen_forward_dynamics_gradient_device_function_call(compute_Minv=use_qdd_Minv_input)
    if self.DEBUG_MODE and not single_call_timing:
        # the reference's DEBUG_MODE prints the same-named intermediates of its NumPy oracle (reference _forward_dynamics_gradient.py:28-46)
        self.gen_add_debug_print_code_lines(["printf(\"Minv\\n\");", "printMat<T,%d,%d>(s_Minv,GRID_MINV_LD);" % (n, n),
                                             "printf(\"qdd\\n\");", "printMat<T,1,%d>(s_qdd,1);" % n,
                                             "printf(\"df/dq\\n\");", "printMat<T,%d,%d>(&s_df_du[0],%d);" % (n, n, n),
                                             "printf(\"df/dqd\\n\");", "printMat<T,%d,%d>(&s_df_du[%d],%d);" % (n, n, n * n, n)], use_thread_group)
    if single_call_timing:
        self.gen_add_end_control_flow()
    if single_call_timing:
        self.gen_kernel_save_result_single_timing("df_du", 2 * n * n, use_thread_group)
    elif not stream:
        self.gen_kernel_save_result("df_du", 2 * n * n, 2 * n * n, use_thread_group)
    if not single_call_timing:
        self.gen_add_end_control_flow()
    self.gen_add_end_function()


def gen_forward_dynamics_gradient_host(self, mode=0):
    single_call_timing = mode == 1
    compute_only = mode == 2
    func_params = ["hd_data is the packaged input and output pointers",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant,",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)",
                   "streams are pointers to HIP streams for async memory transfers (if needed)"]
    name = "forward_dynamics_gradient" + ("_single_timing" if single_call_timing else "") + ("_compute_only" if compute_only else "")
    self.gen_add_func_doc("Computes the gradient of forward dynamics", [], func_params, None)
    self.gen_add_code_line("template <typename T, bool USE_QDD_MINV_FLAG = false>")
    self.gen_add_code_line("__host__")
    self.gen_add_code_line("void " + name + "(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps,")
    self.gen_add_code_line("                      const dim3 block_dimms, const dim3 thread_dimms" + ("" if compute_only else ", hipStream_t *streams") + ") {", True)
    self.gen_add_code_line("int stride_q_qd = 3*NUM_JOINTS;")
    cnt = "" if single_call_timing else "num_timesteps*"
    if not compute_only:
        self.gen_add_code_lines(["// start code with memory transfer",
                                 "gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q_qd*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[0]));",
                                 "if (USE_QDD_MINV_FLAG) {",
                                 "    gpuErrchk(hipMemcpyAsync(hd_data->d_qdd,hd_data->h_qdd,NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[1]));",
                                 "    gpuErrchk(hipMemcpyAsync(hd_data->d_Minv,hd_data->h_Minv,NUM_JOINTS*NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[2]));",
                                 "}",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    kern = "forward_dynamics_gradient_kernel" + ("_single_timing" if single_call_timing else "") + "<T>"
    self.gen_add_code_line("// then call the kernel")
    if single_call_timing:
        self.gen_add_code_line("struct timespec start, end; clock_gettime(CLOCK_MONOTONIC,&start);")
    self.gen_add_code_lines(["if (USE_QDD_MINV_FLAG) {hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms),0,hd_data->d_df_du,hd_data->d_q_qd_u,stride_q_qd,hd_data->d_qdd,hd_data->d_Minv,d_robotModel,gravity,num_timesteps);}",
                             "else {hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, FD_DU_LDS_PER_SOLVE" + ("" if single_call_timing else ", FD_DU_OUT_PER_SOLVE") + "),0,hd_data->d_df_du,hd_data->d_q_qd_u,stride_q_qd,d_robotModel,gravity,num_timesteps);}",
                             "gpuErrchk(hipGetLastError()); gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("clock_gettime(CLOCK_MONOTONIC,&end);")
    if not compute_only:
        self.gen_add_code_lines(["// finally transfer the result back",
                                 "gpuErrchk(hipMemcpy(hd_data->h_df_du,hd_data->d_df_du,NUM_JOINTS*2*NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyDeviceToHost));",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("printf(\"Single Call FD_DU %fus\\n\",time_delta_us_timespec(start,end)/static_cast<double>(num_timesteps));")
    self.gen_add_end_function()


def gen_forward_dynamics_gradient_device_function_call(self, compute_Minv=False):
    if compute_Minv:
        self.gen_add_code_line("forward_dynamics_gradient_device<T>(s_df_du, s_q, s_qd, s_qdd, s_Minv, s_mem, d_robotModel, gravity, lane);")
    else:
        self.gen_add_code_line("forward_dynamics_gradient_device<T>(s_df_du, s_q, s_qd, s_u, s_mem, d_robotModel, gravity, lane);")


def gen_forward_dynamics_gradient(self, use_thread_group=False):
    # first device wrappers
    self.gen_forward_dynamics_gradient_device(use_thread_group, False)
    self.gen_forward_dynamics_gradient_device(use_thread_group, True)
    if getattr(self, "fd_stream_out", False):
        self.gen_forward_dynamics_gradient_stream_device(use_thread_group)
    # then kernels
    self.gen_forward_dynamics_gradient_kernel(use_thread_group, True, True)
    self.gen_forward_dynamics_gradient_kernel(use_thread_group, True, False)
    self.gen_forward_dynamics_gradient_kernel(use_thread_group, False, True)
    self.gen_forward_dynamics_gradient_kernel(use_thread_group, False, False)
    # finally host wrappers
    self.gen_forward_dynamics_gradient_host(0)
    self.gen_forward_dynamics_gradient_host(1)
    self.gen_forward_dynamics_gradient_host(2)
