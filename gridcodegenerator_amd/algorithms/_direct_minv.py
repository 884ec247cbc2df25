"""Analytical inverse of the joint-space inertia matrix, emitter for the HIP/CDNA4 backend.

Mirrors the role of the reference's algorithms/_direct_minv.py (gen_direct_minv_inner :23-453, device :455,
kernel :478, host :527) and follows the reference oracle /root/reference/_test.py:117-226
(test_minv_bpass / test_minv_fpass / test_densify_Minv; Carpentier's analytical M^-1).

Lane mapping (lane j of the solve's lane group):
  * owns COLUMN j of M^-1 (register vector Mcol[i] = Minv[i][j]) and the matching column of every F_i;
    F_i[:,j] is zero unless j is in subtree(i), so zeros propagate structurally and no masks are needed
    (for the oracle's subtree index sets, _test.py:151-168);
  * lanes 0..5 additionally own COLUMN j of every articulated inertia IA_i.  The two-sided update
    IA_p += X^T (IA - U D^-1 U^T) X is done as two one-sided "X^T * 6-vector" products with a 6x6 transpose
    through LDS in between (R[:,c] = X^T (X^T Ia)[c,:]^T, valid because Ia is symmetric), i.e. 2x27 FMAs per
    lane instead of ~324 on one thread;
  * U_i = IA_i[:, S_i] is published through LDS by the lane that owns that column.
The forward sweep (serial over joints in the reference, :371-452) becomes a pre-order walk where each lane
carries its own column of F.
"""


def gen_direct_minv_inner_temp_mem_size(self):
    return 0  # LDS needs are part of the fixed per-solve slice (helpers/_topology_helpers.py: gen_lds_layout)


def gen_direct_minv_inner_function_call(self, use_thread_group=False, updated_var_names=None):
    self.gen_add_code_line("direct_minv_inner<T>(s_Minv, s_X, s_U, s_T, d_robotModel, lane);")
    self.gen_add_sync(use_thread_group)


def gen_direct_minv_inner(self, use_thread_group=False, body_only=False, bwd_hook=None, fwd_hook=None):
    """body_only: emit just the function body (the fused forward_dynamics_inner supplies the header); bwd_hook(idx) is called once
    per backward-sweep joint inside the region that waits for U (independent work placed there hides the LDS round trip);
    fwd_hook() is called after the forward sweep, before M^-1 is published."""
    m = self.model
    n = m.n
    if not body_only:
        self.gen_direct_minv_inner_header()
    self.gen_direct_minv_inner_body(use_thread_group, bwd_hook, fwd_hook)
    if not body_only:
        self.gen_add_end_function()


def gen_direct_minv_inner_header(self):
    self.gen_add_func_doc("Compute the inverse of the mass matrix (dense, symmetric) into LDS",
                          ["follows /root/reference/_test.py:117-226; lane j produces column j (and, by symmetry, row j)",
                           "the caller must grid_wave_sync() before other lanes' entries of s_Minv are read"],
                          ["s_Minv is the n x n output in LDS (symmetric, so row- and column-major coincide)",
                           "s_X is this solve's compact X(q) storage", "s_U is LDS scratch: per joint U (6), 1/D, pad",
                           "s_T is LDS scratch for the 6x6 transpose (row stride 8)",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU",
                           "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void direct_minv_inner(T *s_Minv, const T *s_X, T *s_U, T *s_T, const robotModel<T> *d_robotModel, const int lane) {", True)


def gen_direct_minv_inner_body(self, use_thread_group=False, bwd_hook=None, fwd_hook=None):
    m = self.model
    n = m.n
    bwd_count = [0]
    IA0 = 0 if self.cols_per_lane == 2 else self.lanes_per_solve // 2  # first of the 6 lanes that carry the articulated-inertia columns
    shared = IA0 != 0  # F lanes and IA lanes are disjoint: one X^T product serves both
    if IA0 == 0:
        self.gen_add_code_line("const int cI = lane < 5 ? lane : 5; // articulated-inertia column owned by this lane (lanes 0..5 are live)")
        self.gen_add_code_line("const bool isIA = lane < 6;")
    else:
        self.gen_add_code_line("const int cI = (lane < %d) ? 0 : ((lane > %d) ? 5 : (lane - %d)); // articulated-inertia column owned by this lane (lanes %d..%d are live)" % (IA0, IA0 + 5, IA0, IA0, IA0 + 5))
        self.gen_add_code_line("const bool isIA = (lane >= %d) && (lane < %d);" % (IA0, IA0 + 6))
    self.gen_add_code_line("const T *d_I = &grid_model_constants(static_cast<const T *>(nullptr))[" + str(18 * n) + " + 6*cI]; (void)d_robotModel;")
    self.gen_add_code_line("T Mcol[" + str(n) + "];")
    keep_U_in_regs = n <= 9  # 7 VGPRs per joint; larger robots park U_i, 1/D_i in LDS for the forward sweep
    if keep_U_in_regs:
        self.gen_add_code_line("T Uk[%d][6], Dk[%d]; // U_i and 1/D_i of every joint, kept for the forward sweep" % (n, n))
        self.gen_add_code_line("(void)s_U;")
    self.gen_add_code_line("//")
    self.gen_add_code_line("// backward sweep (post-order): U, D^-1, Minv row updates, F and IA propagation to the parent")
    self.gen_add_code_line("//")

    def pre_b(i):
        self.gen_add_code_line("T IA_%d[6], F_%d[6];" % (i, i))
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { IA_%d[r] = d_I[%d + r]; F_%d[r] = static_cast<T>(0); }" % (i, 36 * i, i))

    def post_b(i):
        s, p = m.S_index[i], m.parent[i]
        tbuf = (m.depth[i] & 1) * 40  # the transpose scratch is double buffered by depth parity
        self.gen_add_code_line("{", True)
        self.gen_add_code_line("// U = IA[:, S] lives in the lane that owns that column: broadcast it inside the lane group (no LDS hand-off, no sync)")
        self.gen_add_code_line("T U[6];")
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { U[r] = grid_group_shfl(IA_%d[r], %d); }" % (i, IA0 + s))
        self.gen_add_code_line("const T Dinv = grid_rcp(U[%d]);" % s)
        if keep_U_in_regs:
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { Uk[%d][r] = U[r]; }" % i)
            self.gen_add_code_line("Dk[%d] = Dinv;" % i)
        else:
            self.gen_add_code_line("if (lane == 0) {", True)
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { s_U[%d + r] = U[r]; }" % (8 * i))
            self.gen_add_code_line("s_U[%d] = Dinv;" % (8 * i + 6))
            self.gen_add_end_control_flow()
        self.gen_add_code_line("const T m = ((lane == %d) ? Dinv : static_cast<T>(0)) - Dinv*F_%d[%d];" % (i, i, s))
        self.gen_add_code_line("Mcol[%d] = m; grid_pin(Mcol[%d]);" % (i, i))
        if bwd_hook is not None:
            saved = self._cur_joint
            bwd_hook(bwd_count[0])
            self._cur_joint = saved
            bwd_count[0] += 1
        if p != -1:
            self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*%d]);" % i)
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { F_%d[r] += U[r]*m; }" % i)
            self.gen_add_code_line("// IA_parent += X^T (IA - U Dinv U^T) X, one column per lane, transposed through LDS")
            self.gen_add_code_line("T Ia[6], Tc[6], Tr[6];")
            self.gen_add_code_line("const T w = Dinv*IA_%d[%d];" % (i, s))
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { Ia[r] = IA_%d[r] - U[r]*w; }" % i)
            if not shared:
                self.gen_add_code_line("grid_xtmul_peq(F_%d, X, F_%d); grid_pin6(F_%d);" % (p, i, p))
                self.gen_add_code_line("grid_xtmul(Tc, X, Ia);")
            else:
                self.gen_add_code_line("// the F lanes and the IA lanes are disjoint: one X^T product serves both")
                self.gen_add_code_line("#pragma unroll")
                self.gen_add_code_line("for (int r = 0; r < 6; r++) { Ia[r] = isIA ? Ia[r] : F_%d[r]; }" % i)
                self.gen_add_code_line("grid_xtmul(Tc, X, Ia);")
                self.gen_add_code_line("#pragma unroll")
                self.gen_add_code_line("for (int r = 0; r < 6; r++) { F_%d[r] += isIA ? static_cast<T>(0) : Tc[r]; }" % p)
                self.gen_add_code_line("grid_pin6(F_%d);" % p)
            self.gen_add_code_line("if (isIA) {", True)
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { s_T[%d + 6*r + cI] = Tc[r]; }" % tbuf)
            self.gen_add_end_control_flow()
            self.gen_add_sync(use_thread_group)
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { Tr[r] = s_T[%d + 6*cI + r]; }" % tbuf)
            self.gen_add_code_line("grid_xtmul_peq(IA_%d, X, Tr); grid_pin6(IA_%d);" % (p, p))
        self.gen_add_end_control_flow()

    self.gen_tree_traversal(pre_b, post_b)
    self.gen_add_sync(use_thread_group)
    self.gen_add_code_line("//")
    self.gen_add_code_line("// forward sweep (pre-order): Minv[i,j] -= Dinv_i U_i^T (X_i F_parent[:,j]);  F_i[:,j] = S_i Minv[i,j] + X_i F_parent[:,j]")
    self.gen_add_code_line("// (entries with j < i are scratch: only the upper triangle is defined, the store below mirrors it)")
    self.gen_add_code_line("//")

    def pre_f(i):
        s, p = m.S_index[i], m.parent[i]
        has_children = len(m.children[i]) > 0
        if p == -1:
            if has_children:
                self.gen_add_code_line("T Ff_%d[6]; grid_zero6(Ff_%d); Ff_%d[%d] = Mcol[%d];" % (i, i, i, s, i))
            return
        self.gen_add_code_line("T Ff_%d[6];" % i)
        self.gen_add_code_line("{", True)
        self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*%d]);" % i)
        self.gen_add_code_line("grid_xmul(Ff_%d, X, Ff_%d);" % (i, p))
        if keep_U_in_regs:
            self.gen_add_code_line("Mcol[%d] -= Dk[%d]*grid_dot6(Uk[%d], Ff_%d); grid_pin(Mcol[%d]);" % (i, i, i, i, i))
        else:
            self.gen_add_code_line("{ T U[6];")
            self.gen_add_code_line("  #pragma unroll")
            self.gen_add_code_line("  for (int r = 0; r < 6; r++) { U[r] = s_U[%d + r]; }" % (8 * i))
            self.gen_add_code_line("  Mcol[%d] -= s_U[%d]*grid_dot6(U, Ff_%d); grid_pin(Mcol[%d]); }" % (i, 8 * i + 6, i, i))
        if has_children:
            self.gen_add_code_line("Ff_%d[%d] += Mcol[%d];" % (i, s, i))
        self.gen_add_end_control_flow()

    def post_f(i):
        pass

    self.gen_tree_traversal(pre_f, post_f)
    if fwd_hook is not None:
        fwd_hook()
    self.gen_add_code_line("// publish: lane j writes the upper-triangle entries of its column and their mirror images")
    self.gen_add_code_line("if (lane < %d) {" % n, True)
    for i in range(n):
        self.gen_add_code_line("if (lane >= %d) { s_Minv[%d + lane] = Mcol[%d]; s_Minv[lane*%d + %d] = Mcol[%d]; }" % (i, i * self.minv_ld, i, self.minv_ld, i, i))
    self.gen_add_end_control_flow()


def gen_direct_minv_device(self, use_thread_group=False):
    self.gen_add_func_doc("Compute the inverse of the mass matrix: X(q) update + direct_minv_inner (lane-group cooperative)",
                          ["all lanes of the solve's lane group must call it; s_Minv (dense, symmetric) is visible to the group on return"],
                          ["s_Minv is the n x n output in LDS", "s_q is the vector of joint positions in LDS",
                           "s_work is this solve's LDS workspace of GRID_LDS_PER_SOLVE elements", "d_robotModel is the pointer to the initialized model specific helpers on the GPU",
                           "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void direct_minv_device(T *s_Minv, const T *s_q, T *s_work, const robotModel<T> *d_robotModel, const int lane, const int off_sp = GRID_OFF_SP) {", True)
    self.gen_add_code_line("T *s_X = &s_work[GRID_OFF_X]; T *s_U = &s_work[GRID_OFF_U]; T *s_T = &s_work[GRID_OFF_T]; (void)off_sp; // (off_sp: path-axis scratch of branch-frame robots inside s_work; the stand-alone kernel carves a compact slice)")
    self.gen_load_update_XImats_helpers_function_call(use_thread_group)
    if self.tip_frame:  # serial revolute chains: M from the tip-frame composites, factored in registers, one unit-vector solve per lane
        self.gen_add_code_line("(void)s_T;")
        self.gen_add_code_line("direct_minv_inner_tip<T>(s_Minv, s_X, s_U, d_robotModel, lane);")
        self.gen_add_sync(use_thread_group)
        self.gen_add_end_function()
        return
    if getattr(self, "branch_components", False):  # branched revolute robots: tree-sparse factorisation of M from the branch-frame composites, one unit-vector solve per lane
        self.gen_add_code_line("(void)s_T; (void)s_U;")
        self.gen_add_code_line("direct_minv_inner_branch<T>(s_Minv, s_X, &s_work[off_sp], d_robotModel, lane);")
        self.gen_add_sync(use_thread_group)
        self.gen_add_end_function()
        return
    self.gen_direct_minv_inner_function_call(use_thread_group)
    self.gen_add_end_function()


def gen_direct_minv_kernel(self, use_thread_group=False, single_call_timing=False):
    n = self.model.n
    func_params = ["d_Minv is the output: upper triangle of M^-1, column-major (d_Minv[k*n*n + col*n + row], entries below the diagonal are 0)",
                   "d_q is the vector of joint positions", "stride_q is the stride between each q",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)"]
    func_def = "void direct_minv_kernel(T *d_Minv, const T *d_q, const int stride_q, const robotModel<T> *d_robotModel, const int NUM_TIMESTEPS) {"
    if single_call_timing:
        func_def = func_def.replace("kernel(", "kernel_single_timing(")
    self.gen_add_func_doc("Compute the inverse of the mass matrix", ["Outputs a SYMMETRIC_UPPER triangular matrix for Minv (as the reference does)"], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__global__ GRID_LAUNCH_BOUNDS")
    self.gen_add_code_line(func_def, True)
    self.gen_kernel_prologue("MINV_LDS_PER_SOLVE")
    self.gen_add_code_lines(["T *s_q = &s_mem[GRID_OFF_IN];",
                             "T *s_Minv = &s_mem[MINV_OFF_MINV]; T *s_out = &s_out_all[grp*%d]; (void)s_out;" % (n * n)])
    if single_call_timing:
        self.gen_add_code_line("const int k = 0; const int kc = 0; const bool valid = (blockIdx.x + blockIdx.y == 0) && (grp == 0); const int lane = lane_id; const int NUM_TIMESTEPS_OUT = 1;")
        self.gen_add_code_line("if (!valid) {return;}")
    else:
        self.gen_add_parallel_loop("k", "NUM_TIMESTEPS", use_thread_group, block_level=True)
    self.gen_kernel_load_inputs("q", "stride_q", n, use_thread_group)
    if single_call_timing:
        self.gen_add_code_line("for (int rep = 0; rep < NUM_TIMESTEPS; rep++){", True)
    self.gen_add_code_line("// compute")
    self.gen_add_code_line("direct_minv_device<T>(s_Minv, s_q, s_mem, d_robotModel, lane, MINV_OFF_SP);")
    if single_call_timing:
        self.gen_add_end_control_flow()
    direct_out = self.gen_lds_layout()["KERNELS"]["MINV"]["OUT"] == 0 and not single_call_timing
    if direct_out:
        self.gen_add_code_line("// upper triangle only in the output record; compact slice: no second staging copy, every lane gathers 16-byte pieces of the record from s_Minv")
        self.gen_add_code_line("if (valid) {", True)
        self.gen_add_code_line("T *dst = &d_Minv[static_cast<size_t>(kc)*%d];" % (n * n))
        self.gen_add_code_line("for (int e = 4*lane; e + 3 < %d; e += 4*GRID_LANES_PER_SOLVE) {" % (n * n), True)
        self.gen_add_code_line("T tmp[4];")
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 4; r++) { const int ind = e + r; const int row = ind %% %d; const int col = ind / %d; tmp[r] = (row <= col) ? s_Minv[col*%d + row] : static_cast<T>(0); }" % (n, n, self.minv_ld))
        self.gen_add_code_line("grid_store4(dst + e, tmp);")
        self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
        self.gen_add_sync(use_thread_group)
    else:
        self.gen_add_code_line("// upper triangle only in the output record")
        self.gen_add_parallel_loop("ind", str(n * n), use_thread_group)
        self.gen_add_code_line("const int row = ind %% %d; const int col = ind / %d;" % (n, n))
        self.gen_add_code_line("s_out[ind] = (row <= col) ? s_Minv[col*%d + row] : static_cast<T>(0);" % self.minv_ld)
        self.gen_add_end_control_flow()
        if single_call_timing:
            self.gen_kernel_save_result_single_timing("Minv", n * n, use_thread_group, "s_out")
        else:
            self.gen_kernel_save_result("Minv", n * n, n * n, use_thread_group, "s_out")
    if not single_call_timing:
        self.gen_add_end_control_flow()
    self.gen_add_end_function()


def gen_direct_minv_host(self, mode=0):
    single_call_timing = mode == 1
    compute_only = mode == 2
    func_params = ["hd_data is the packaged input and output pointers",
                   "d_robotModel is the pointer to the initialized model specific helpers on the GPU (XImats, topology_helpers, etc.)",
                   "num_timesteps is the length of the trajectory points we need to compute over (or overloaded as test_iters for timing)",
                   "streams are pointers to HIP streams for async memory transfers (if needed)"]
    name = "direct_minv" + ("_single_timing" if single_call_timing else "") + ("_compute_only" if compute_only else "")
    self.gen_add_func_doc("Compute the inverse of the mass matrix", [], func_params, None)
    self.gen_add_code_line("template <typename T, bool USE_COMPRESSED_MEM = false>")
    self.gen_add_code_line("__host__")
    self.gen_add_code_line("void " + name + "(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const int num_timesteps,")
    self.gen_add_code_line("                      const dim3 block_dimms, const dim3 thread_dimms" + ("" if compute_only else ", hipStream_t *streams") + ") {", True)
    self.gen_add_code_line("int stride_q = USE_COMPRESSED_MEM ? NUM_JOINTS: 3*NUM_JOINTS;")
    cnt = "" if single_call_timing else "num_timesteps*"
    if not compute_only:
        self.gen_add_code_lines(["// start code with memory transfer",
                                 "if (USE_COMPRESSED_MEM) {gpuErrchk(hipMemcpyAsync(hd_data->d_q,hd_data->h_q,stride_q*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[0]));}",
                                 "else {gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q*" + cnt + "sizeof(T),hipMemcpyHostToDevice,streams[0]));}",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    kern = "direct_minv_kernel" + ("_single_timing" if single_call_timing else "") + "<T>"
    self.gen_add_code_line("// then call the kernel")
    self.gen_add_code_line("const T *d_in = USE_COMPRESSED_MEM ? hd_data->d_q : hd_data->d_q_qd_u;")
    if single_call_timing:
        self.gen_add_code_line("struct timespec start, end; clock_gettime(CLOCK_MONOTONIC,&start);")
    self.gen_add_code_lines(["hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, MINV_LDS_PER_SOLVE, " + ("GRID_OUT_PER_SOLVE" if single_call_timing else "MINV_OUT_PER_SOLVE") + "),0,hd_data->d_Minv,d_in,stride_q,d_robotModel,num_timesteps);",
                             "gpuErrchk(hipGetLastError()); gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("clock_gettime(CLOCK_MONOTONIC,&end);")
    if not compute_only:
        self.gen_add_code_lines(["// finally transfer the result back",
                                 "gpuErrchk(hipMemcpy(hd_data->h_Minv,hd_data->d_Minv,NUM_JOINTS*NUM_JOINTS*" + cnt + "sizeof(T),hipMemcpyDeviceToHost));",
                                 "gpuErrchk(hipDeviceSynchronize());"])
    if single_call_timing:
        self.gen_add_code_line("printf(\"Single Call Minv %fus\\n\",time_delta_us_timespec(start,end)/static_cast<double>(num_timesteps));")
    self.gen_add_end_function()


def gen_direct_minv(self, use_thread_group=False):
    self.gen_direct_minv_inner(use_thread_group)
    self.gen_direct_minv_device(use_thread_group)
    self.gen_direct_minv_kernel(use_thread_group, True)
    self.gen_direct_minv_kernel(use_thread_group, False)
    for mode in (0, 1, 2):
        self.gen_direct_minv_host(mode)
