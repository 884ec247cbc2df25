"""RNEA (inverse dynamics) emitter for the HIP/CDNA4 backend.

Mirrors the role of the reference's algorithms/_inverse_dynamics.py (gen_inverse_dynamics_inner :33-321, device :328,
kernel :371, host :440) and follows the mathematics of the reference oracle /root/reference/_test.py:5-115.

Design: the RNEA recursion has no column parallelism, so every lane of the solve's lane group evaluates it redundantly
in registers (wave-uniform work costs the same issue slots whether 1 or G lanes do it), walking the kinematic tree in
depth-first order so that only O(depth) link vectors are live.  c = S^T f + damping*qd follows the oracle (the
reference's emitted CUDA has no damping term).
"""


def gen_tree_traversal(self, pre_fn, post_fn):
    """Depth-first walk of the kinematic tree; each joint opens a C++ scope that nests its subtree, so a parent's
    register vectors stay visible to its children and die when its scope closes."""
    m = self.model

    def visit(i):
        self.gen_add_code_line("{ // joint " + str(i) + (" (root)" if m.parent[i] == -1 else " (parent " + str(m.parent[i]) + ")"), True)
        self._cur_joint = i
        pre_fn(i)
        self._cur_joint = None
        for ch in m.children[i]:
            visit(ch)
        self._cur_joint = i
        post_fn(i)
        self._cur_joint = None
        self.gen_add_end_control_flow()

    for r in m.roots:
        visit(r)


def gen_inverse_dynamics_inner_temp_mem_size(self):
    return 0  # register resident


def gen_inverse_dynamics_inner_function_call(self, use_thread_group=False, compute_c=True, use_qdd_input=False, updated_var_names=None):
    var = dict(c_name="c", s_qd_name="s_qd", s_qdd_name="s_qdd", s_X_name="s_X")
    if updated_var_names is not None:
        var.update(updated_var_names)
    call = "inverse_dynamics_inner<T>(" + var["c_name"] + ", " + var["s_qd_name"] + ", "
    if use_qdd_input:
        call += var["s_qdd_name"] + ", "
    call += var["s_X_name"] + ", gravity);"
    self.gen_add_code_line(call)


def gen_inverse_dynamics_inner(self, use_thread_group=False, use_qdd_input=False):
    m = self.model
    n = m.n
    params = ["c is the register vector receiving the bias force / inverse dynamics torque (identical in every lane)",
              "s_qd is the vector of joint velocities in LDS"]
    if use_qdd_input:
        params.append("s_qdd is the vector of joint accelerations in LDS")
    params += ["s_X is this solve's compact X(q) storage (see load_update_XImats_helpers)", "gravity is the gravity constant (positive, 9.81)"]
    self.gen_add_func_doc("Compute the RNEA (Recursive Newton-Euler Algorithm)" + ("" if use_qdd_input else " with qdd = 0"),
                          ["wave-uniform: every lane of the solve's lane group computes the same values in registers",
                           "follows /root/reference/_test.py:5-115 (test_rnea_fpass/test_rnea_bpass), incl. velocity damping"], params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void inverse_dynamics_inner(T (&c)[" + str(n) + "], const T *s_qd, " + ("const T *s_qdd, " if use_qdd_input else "") +
                           "const T *s_X, const T gravity) {", True)

    dbg = bool(getattr(self, "DEBUG_MODE", False))
    if dbg:
        # the reference's DEBUG_MODE dumps (reference algorithms/_inverse_dynamics.py:73-83,137-144,238-252,298-305): same names as its NumPy oracle prints
        # (_test.py:33-37,51-55), from the thread that owns joint 0 of the first solve of the launch; X(q) is kept in compact form here: E | B instead of 6 x 6
        self.gen_add_code_line("const bool dbg = (threadIdx.x + threadIdx.y*blockDim.x == 0) && (blockIdx.x + blockIdx.y == 0);")
        self.gen_add_code_line("if (dbg) {", True)
        self.gen_add_code_line("printf(\"qd\\n\"); printMat<T,1,%d>(s_qd,1);" % n)
        if use_qdd_input:
            self.gen_add_code_line("printf(\"qdd\\n\"); printMat<T,1,%d>(s_qdd,1);" % n)
        self.gen_add_code_line("for (int i = 0; i < %d; i++){printf(\"X[%%d] (compact: E rows, then B rows)\\n\",i); printMat<T,3,6>(&s_X[GRID_X_STRIDE*i],3);}" % n)
        self.gen_add_end_control_flow()

    def pre(i):
        s, p = m.S_index[i], m.parent[i]
        self.gen_add_code_line("const T qd_%d = s_qd[%d];" % (i, i))
        self.gen_add_code_line("T v_%d[6], a_%d[6], f_%d[6];" % (i, i, i))
        self.gen_add_code_line("{", True)
        self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*%d]);" % i)
        if p == -1:
            self.gen_add_code_line("grid_zero6(v_%d); v_%d[%d] = qd_%d;" % (i, i, s, i))
            self.gen_add_code_line("grid_zero6(a_%d); a_%d[3] = X[2]*gravity; a_%d[4] = X[5]*gravity; a_%d[5] = X[8]*gravity; // X*[0,0,0,0,0,g]" % (i, i, i, i))
        else:
            self.gen_add_code_line("grid_xmul(v_%d, X, v_%d); v_%d[%d] += qd_%d;" % (i, p, i, s, i))
            self.gen_add_code_line("grid_xmul(a_%d, X, a_%d); grid_mxS_peq<T,%d>(a_%d, v_%d, qd_%d);" % (i, p, s, i, i, i))
        if use_qdd_input:
            self.gen_add_code_line("a_%d[%d] += s_qdd[%d];" % (i, s, i))
        self.gen_add_code_line("T Iv[6]; grid_imul_%d(Iv, v_%d); grid_imul_%d(f_%d, a_%d); grid_fxv_peq(f_%d, v_%d, Iv); grid_pin6(f_%d);" % (i, i, i, i, i, i, i, i))
        if dbg:
            self.gen_add_code_line("if (dbg) { printf(\"s_v[%d]\\n\"); printMat<T,1,6>(v_%d,1); printf(\"s_a[%d]\\n\"); printMat<T,1,6>(a_%d,1); printf(\"s_f[%d] (forward pass)\\n\"); printMat<T,1,6>(f_%d,1); }" % (i, i, i, i, i, i))
        self.gen_add_end_control_flow()

    def post(i):
        s, p = m.S_index[i], m.parent[i]
        damp = m.damping[i]
        self.gen_add_code_line("c[%d] = f_%d[%d]%s; grid_pin(c[%d]);" % (i, i, s, (" + static_cast<T>(" + repr(float(damp)) + ")*qd_" + str(i)) if damp != 0.0 else "", i))
        if dbg:
            self.gen_add_code_line("if (dbg) { printf(\"s_f[%d] (after the backward pass)\\n\"); printMat<T,1,6>(f_%d,1); printf(\"c[%d] = %%.4f\\n\", static_cast<double>(c[%d])); }" % (i, i, i, i))
        if p != -1:
            self.gen_add_code_line("{ T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*%d]); grid_xtmul_peq(f_%d, X, f_%d); grid_pin6(f_%d); }" % (i, p, i, p))

    self.gen_tree_traversal(pre, post)
    self.gen_add_end_function()


def gen_inverse_dynamics_device(self, use_thread_group=False, use_qdd_input=False):
    n = self.model.n
    params = ["s_c is the output vector of torques in LDS", "s_q is the vector of joint positions in LDS", "s_qd is the vector of joint velocities in LDS"]
    if use_qdd_input:
        params.append("s_qdd is the vector of joint accelerations in LDS")
    params += ["s_work is this solve's LDS workspace of GRID_LDS_PER_SOLVE elements", "d_robotModel is the pointer to the initialized model specific helpers on the GPU",
               "gravity is the gravity constant", "lane is the caller's lane index inside the solve's lane group"]
    self.gen_add_func_doc("Compute the RNEA (Recursive Newton-Euler Algorithm): X(q) update + inverse_dynamics_inner (lane-group cooperative)",
                          ["all lanes of the solve's lane group must call it; s_c is visible to the group on return"], params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void inverse_dynamics_device(T *s_c, const T *s_q, const T *s_qd, " + ("const T *s_qdd, " if use_qdd_input else "") +
                           "T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane, const int off_sp = GRID_OFF_SP) {", True)
    self.gen_add_code_line("T *s_X = &s_work[GRID_OFF_X]; (void)off_sp; // (off_sp: where the path-axis scratch of branch-frame robots starts inside s_work; the stand-alone kernel carves a compact slice)")
    self.gen_load_update_XImats_helpers_function_call(use_thread_group)
    if self.tip_frame:  # serial revolute chains: RNEA in the tip link's frame, lane j produces c[j]
        self.gen_add_code_line("inverse_dynamics_inner_tip<T>(s_c, s_qd, %s, s_X, gravity, d_robotModel, lane);" % ("s_qdd" if use_qdd_input else "static_cast<const T *>(nullptr)"))
        self.gen_add_sync(use_thread_group)
        self.gen_add_end_function()
        return
    if getattr(self, "branch_components", False):  # branched revolute robots: RNEA with every branch in its tip link's frame, the lane of joint j produces c[j]
        self.gen_add_code_line("inverse_dynamics_inner_branch<T>(s_c, s_qd, %s, s_X, &s_work[off_sp], d_robotModel, gravity, lane);" % ("s_qdd" if use_qdd_input else "static_cast<const T *>(nullptr)"))
        self.gen_add_sync(use_thread_group)
        self.gen_add_end_function()
        return
    self.gen_add_code_line("T c[%d];" % n)
    self.gen_inverse_dynamics_inner_function_call(use_thread_group, True, use_qdd_input)
    self.gen_add_code_line("if (lane == 0) {", True)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int i = 0; i < %d; i++) { s_c[i] = c[i]; }" % n)
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)
    self.gen_add_end_function()


def gen_inverse_dynamics_kernel(self, use_thread_group=False, use_qdd_input=False, single_call_timing=False):
    n = self.model.n
    func_params = ["d_c is the vector of output torques", "d_q_dq is the vector of joint positions and velocities",
                   "stride_q_qd is the stride between each q, qd",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)"]
    func_def = "void inverse_dynamics_kernel(T *d_c, const T *d_q_qd, const int stride_q_qd, "
    if use_qdd_input:
        func_def += "const T *d_qdd, "
        func_params.insert(-3, "d_qdd is the vector of joint accelerations")
    func_def += "const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {"
    if single_call_timing:
        func_def = func_def.replace("kernel(", "kernel_single_timing(")
    self.gen_add_func_doc("Compute the RNEA (Recursive Newton-Euler Algorithm)", [], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__global__ GRID_LAUNCH_BOUNDS")
    self.gen_add_code_line(func_def, True)
    self.gen_kernel_prologue("ID_LDS_PER_SOLVE")
    self.gen_add_code_lines(["T *s_q_qd = &s_mem[GRID_OFF_IN]; T *s_q = s_q_qd; T *s_qd = &s_q_qd[%d]; T *s_qdd = &s_q_qd[%d];" % (n, 2 * n),
                             "T *s_c = &s_out_all[grp*%d];" % n])
    if single_call_timing:
        self.gen_add_code_line("const int k = 0; const int kc = 0; const bool valid = (blockIdx.x + blockIdx.y == 0) && (grp == 0); const int lane = lane_id; const int NUM_TIMESTEPS_OUT = 1;")
        self.gen_add_code_line("if (!valid) {return;}")
    else:
        self.gen_add_parallel_loop("k", "NUM_TIMESTEPS", use_thread_group, block_level=True)
    if use_qdd_input:
        self.gen_kernel_load_inputs("q_qd", "stride_q_qd", 2 * n, use_thread_group, "qdd", n, n)
    else:
        self.gen_kernel_load_inputs("q_qd", "stride_q_qd", 2 * n, use_thread_group)
    if single_call_timing:
        self.gen_add_code_line("for (int rep = 0; rep < NUM_TIMESTEPS; rep++){", True)
    self.gen_add_code_line("// compute")
    self.gen_add_code_line("inverse_dynamics_device<T>(s_c, s_q, s_qd, " + ("s_qdd, " if use_qdd_input else "") + "s_mem, d_robotModel, gravity, lane, ID_OFF_SP);")
    if single_call_timing:
        self.gen_add_end_control_flow()
    if single_call_timing:
        self.gen_kernel_save_result_single_timing("c", n, use_thread_group)
    else:
        self.gen_kernel_save_result("c", n, n, use_thread_group)
    if not single_call_timing:
        self.gen_add_end_control_flow()
    self.gen_add_end_function()


def gen_inverse_dynamics_host(self, mode=0):
    single_call_timing = mode == 1
    compute_only = mode == 2
    func_params = ["hd_data is the packaged input and output pointers",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "gravity is the gravity constant,",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)",
                   "streams are pointers to HIP streams for async memory transfers (if needed)"]
    name = "inverse_dynamics" + ("_single_timing" if single_call_timing else "") + ("_compute_only" if compute_only else "")
    self.gen_add_func_doc("Compute the RNEA (Recursive Newton-Euler Algorithm)", [], func_params, None)
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
    kern = "inverse_dynamics_kernel" + ("_single_timing" if single_call_timing else "") + "<T>"
    self.gen_add_code_line("// then call the kernel")
    self.gen_add_code_line("const T *d_in = USE_COMPRESSED_MEM ? hd_data->d_q_qd : hd_data->d_q_qd_u;")
    if single_call_timing:
        self.gen_add_code_line("struct timespec start, end; clock_gettime(CLOCK_MONOTONIC,&start);")
    self.gen_add_code_lines(["if (USE_QDD_FLAG) {hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, ID_LDS_PER_SOLVE, ID_OUT_PER_SOLVE),0,hd_data->d_c,d_in,stride_q_qd,hd_data->d_qdd,d_robotModel,gravity,num_timesteps);}",
                             "else {hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, ID_LDS_PER_SOLVE, ID_OUT_PER_SOLVE),0,hd_data->d_c,d_in,stride_q_qd,d_robotModel,gravity,num_timesteps);}",
                             "gpuErrchk(hipGetLastError()); gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("clock_gettime(CLOCK_MONOTONIC,&end);")
    if not compute_only:
        self.gen_add_code_lines(["// finally transfer the result back",
                                 "gpuErrchk(hipMemcpy(hd_data->h_c,hd_data->d_c,NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyDeviceToHost));",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("printf(\"Single Call ID %fus\\n\",time_delta_us_timespec(start,end)/static_cast<double>(num_timesteps));")
    self.gen_add_end_function()


def gen_inverse_dynamics(self, use_thread_group=False):
    # inner (register) functions, both variants
    self.gen_inverse_dynamics_inner(use_thread_group, use_qdd_input=False)
    self.gen_inverse_dynamics_inner(use_thread_group, use_qdd_input=True)
    # device wrappers
    self.gen_inverse_dynamics_device(use_thread_group, use_qdd_input=False)
    self.gen_inverse_dynamics_device(use_thread_group, use_qdd_input=True)
    # kernels
    for use_qdd in (True, False):
        for timing in (True, False):
            self.gen_inverse_dynamics_kernel(use_thread_group, use_qdd, timing)
    # host wrappers
    for mode in (0, 1, 2):
        self.gen_inverse_dynamics_host(mode)
