"""Model constants, LDS layout and the X(q) update for the HIP/CDNA4 backend.

Replaces the hot-path parts of the reference's helpers/_topology_helpers.py (gen_get_XI_size :4, gen_init_XImats :27,
gen_load_update_XImats_helpers :155-331, gen_init_topology_helpers :544, gen_init_robotModel :715).

Differences by design (documented in DESIGN.md):
  * the generator classifies constant / q-dependent entries numerically (X(q) = X_J(q) * X(0)), no sympy
    ``is_constant()`` per entry (the reference spends 50-120 s there for 12-30 DoF robots);
  * the device constant table d_XImats holds, per joint, the compact tree transform [E_T(9) | B_T(9)] followed by the
    column-major 6x6 inertias; per solve only the 18 distinct entries of X(q) live in LDS (reference: 72 floats/joint);
  * the X update is done by lane j for joint j in parallel (reference: serial on thread 0, :221-265) and evaluates
    sin/cos in T (reference: double literals then cast).
"""
import numpy as np


def _pad4(x):
    return (x + 3) // 4 * 4


def gen_lds_layout(self):
    """Per-solve LDS slice (in units of T).  The slice stride is made an odd multiple of 4 dwords so that the
    lane groups of one wave (which read the *same offset* of *different* slices with broadcast ds_reads) land in
    different LDS banks."""
    n = self.model.n
    off = {}
    cur = 0
    t_size = 80
    if getattr(self, "tip_frame", False):  # the tip-frame path keeps 16 (20 with a base-origin family) values per joint in the U|T scratch (M goes into the M^-1 slot)
        t_size = max(t_size, _pad4(self.tip_rec * n - (_pad4(18 * n) if self.reuse_rnea else 8 * n)))
    for name, size in (("IN", _pad4(4 * n)),          # q | qd | u  (| spare: qdd input for the qdd/Minv overloads)
                       ("X", 20 * n),                 # compact X(q): 18 of every 20
                       ("U", _pad4(18 * n) if self.reuse_rnea else 8 * n),  # U_i (6), 1/D_i, pad (deep/large robots); or v, I v, fx(v) I v of RNEA(qdd=0) kept for the gradient walk
                       ("T", t_size),                 # 6x6 transpose scratch (row stride 6, 40 per buffer), double buffered by tree-depth parity
                       ("MINV", n * _pad4(n)),        # dense symmetric M^-1, leading dimension padded to a multiple of 4 (16-byte aligned rows -> ds_read_b128)
                       ("QDD", _pad4(n)),
                       ("F", 0 if self.register_walk else _pad4(6 * n)),                # wave-uniform link forces parked between the two sweeps of the gradient walk
                       ("J", 0 if self.register_walk else 2 * _pad4(6 * n)),      # velocity Jacobian columns of the current link (published by the d/dqd lanes), double buffered
                       ("SP", 0)):  # branch-frame path of the component kernels: joint axes along the root path of every branch (placed below)
        off[name] = cur
        cur += size
    if getattr(self, "branch_components", False):
        need = _pad4(self.branch_plan["sp_size"])
        if cur - off["F"] >= need:
            off["SP"] = off["F"]  # nothing on this path uses the F | J scratch of the column walk
        else:
            off["SP"] = cur
            cur += need
    if (cur // 4) % 2 == 0:
        cur += 4
    off["TOTAL"] = cur
    # the forward-dynamics-gradient kernel (u input) of branch-frame robots needs much less: IN | X | path axes | qdd  (M, its factors and the
    # branch hand-over records re-use the X(q) storage); fewer bytes per solve = more resident waves
    off["FD_SP"] = off["FD_QDD"] = 0
    off["FD_TOTAL"] = cur
    if getattr(self, "branch_frame", False):
        P = self.branch_plan
        from ..algorithms._branch_frame_gradient import branch_spare_in_image
        spare_in_image = branch_spare_in_image(self, 2 * n * n)  # (the spare path records of this kernel live in the head of its result image)
        off["FD_SP"] = off["U"]
        off["FD_QDD"] = off["IN"] + 3 * n  # the qdd-input slot of the other overloads: unused by this kernel
        tot = off["FD_SP"] + _pad4(P["sp_size"] - (6 * P["D"] if spare_in_image else 0)) + int(self.tuning["lds_pad"]) // 4 * 4
        # lane groups of one wave read the same offset of different slices: with two groups per wave any stride that is not a multiple of
        # the 64 banks keeps them apart; more groups want the odd multiple of 4 dwords
        if (tot % 64 == 0) if 64 // self.lanes_per_solve <= 2 else ((tot // 4) % 2 == 0):
            tot += 4
        if tot > cur:
            off["TOTAL"] = cur = tot + (4 if ((tot // 4) % 2 == 0) else 0)
        off["FD_TOTAL"] = tot
    off["OUT_PER_SOLVE"] = _pad4(n * n if self.tuning["out_half"] else 2 * n * n)  # output staging, kept behind all slices (contiguous across the lane groups of a wave)
    # forward_dynamics_gradient kernel of LDS-capacity-bound branch-frame robots: stage one half of the record at a time if that makes more waves
    # resident on a CU (160 KB, allocation granule 512 B, at most 8 waves at 256 VGPRs)
    off["FD_OUT_PER_SOLVE"] = off["OUT_PER_SOLVE"]
    off["FD_YPARK"] = 0
    want = self.tuning["stream_out"]
    self.fd_stream_out = False
    if getattr(self, "branch_frame", False) and want is not False and (n * n) % 4 == 0 and int(self.tuning["debug_stop"]) == 0 and not self.tuning["out_half"]:
        from ..algorithms._branch_frame_gradient import branch_spare_in_image
        if branch_spare_in_image(self, 2 * n * n) != branch_spare_in_image(self, n * n):
            # (the slice above was sized for spare path records inside a full image; a half image of this small robot cannot take them)
            if want is True:
                raise NotImplementedError("stream_out: the half image of this robot is too small for the spare path records of the frame chain")
            want = False
    if getattr(self, "branch_frame", False) and want is not False and (n * n) % 4 == 0 and int(self.tuning["debug_stop"]) == 0 and not self.tuning["out_half"]:
        spw = 64 // self.lanes_per_solve
        park = _pad4(n * max(len(s_) for s_ in self.branch_plan["shapes"]))  # every lane's dc/dqd column, rows of its component
        tot2 = off["FD_TOTAL"] + park
        if (tot2 % 64 == 0) if spw <= 2 else ((tot2 // 4) % 2 == 0):
            tot2 += 4
        waves = lambda slice_, out: min(8, (160 * 1024) // (-(-(spw * (slice_ + out) * 4) // 512) * 512))
        if want is True or (off["FD_TOTAL"] < off["TOTAL"] and waves(tot2, n * n) > waves(off["FD_TOTAL"], off["OUT_PER_SOLVE"])):
            self.fd_stream_out = True
            off["FD_YPARK"] = off["FD_TOTAL"]
            off["FD_TOTAL"] = tot2
            if tot2 > off["TOTAL"]:
                off["TOTAL"] = tot2 + (4 if ((tot2 // 4) % 2 == 0) else 0)
            off["FD_OUT_PER_SOLVE"] = n * n
    # stand-alone kernels (inverse dynamics, its gradient, M^-1, forward dynamics): robots on the branch-frame path need only IN | X | path axes
    # (| M^-1) of the general slice, and their staging is the kernel's own record - fewer bytes per solve = more resident waves
    # (30-DoF humanoid, 16 384 solves: M^-1 107 -> see DESIGN.md section 6b)
    ld = (n + 3) // 4 * 4
    generic = {"LDS": off["TOTAL"], "OUT": off["OUT_PER_SOLVE"], "SP": off["SP"], "MINV": off["MINV"], "compact": False}
    off["KERNELS"] = {k: dict(generic) for k in ("ID", "ID_DU", "MINV", "FD")}
    # aba touches only IN | X | U | T (a prefix of the general slice) and stages n values
    spw_ = 64 // self.lanes_per_solve
    aba = off["MINV"] + (4 if ((off["MINV"] % 64 == 0) if spw_ <= 2 else ((off["MINV"] // 4) % 2 == 0)) else 0)
    off["KERNELS"]["ABA"] = {"LDS": aba, "OUT": _pad4(n), "SP": off["SP"], "MINV": off["MINV"], "compact": aba + _pad4(n) < (off["TOTAL"] + off["OUT_PER_SOLVE"]) // 2}
    if getattr(self, "branch_components", False) and not self.tuning["out_half"]:
        from ..algorithms._branch_frame_gradient import branch_spare_in_image
        P = self.branch_plan
        spw = 64 // self.lanes_per_solve
        fix = lambda t: t + 4 if ((t % 64 == 0) if spw <= 2 else ((t // 4) % 2 == 0)) else t
        sp0 = off["U"]
        sp_len = lambda image: _pad4(P["sp_size"] - (6 * P["D"] if (image and branch_spare_in_image(self, image)) else 0))
        off["KERNELS"]["ID"] = {"LDS": fix(sp0 + sp_len(0)), "OUT": _pad4(n), "SP": sp0, "MINV": off["MINV"], "compact": True}
        off["KERNELS"]["FD"] = dict(off["KERNELS"]["ID"])
        off["KERNELS"]["ID_DU"] = {"LDS": fix(sp0 + sp_len(2 * n * n)), "OUT": _pad4(2 * n * n), "SP": sp0, "MINV": off["MINV"], "compact": True}
        mv = sp0 + sp_len(n * ld)
        # (n^2 a multiple of 4: the kernel gathers its output record straight from s_Minv with 16-byte stores, no second staging copy)
        off["KERNELS"]["MINV"] = {"LDS": fix(mv + n * ld), "OUT": 0 if (n * n) % 4 == 0 else _pad4(n * n), "SP": sp0, "MINV": mv, "compact": True}
    return off


def gen_get_XI_size(self, include_base_inertia=False, include_homogenous_transforms=False):
    """Floats in the device constant table: 18 (compact X_tree) + 36 (inertia) per joint."""
    return 54 * self.model.n


def gen_topology_helpers_size(self):
    return 2 * self.model.n


def gen_model_constant_table(self):
    """Same numbers as init_XImats() but baked into the code object: kernels read them with ONE global load (address known at
    load time) instead of chasing d_robotModel -> d_XImats -> value (three dependent loads at the head of every kernel,
    measured ~0.8 us of a 19 us launch).  d_robotModel stays in every signature for source compatibility and its device
    copy is still filled by init_XImats() for downstream code that reads it."""
    m = self.model
    n = m.n
    vals = []
    for i in range(n):
        XT = m.X_tree[i]
        vals += [XT[r, c] for r in range(3) for c in range(3)] + [XT[3 + r, c] for r in range(3) for c in range(3)]
    for i in range(n):
        vals += [m.I[i][row, col] for col in range(6) for row in range(6)]
    if getattr(self, "tip_frame", False):  # one 12-float row of link constants per lane of the lane group (tip-frame gradient path)
        vals += self.gen_tip_frame_link_constants()
    if getattr(self, "branch_frame", False):  # per-lane rows of the branch-frame gradient path
        assert len(vals) == self.branch_tab_offset
        vals += self.gen_branch_frame_constants()
    if self.gen_idsva_so_mode() == "tree":  # per-lane rows of the tree form of the second-order kernels (algorithms/_idsva_so.py)
        fl, it, K, _ = self.gen_idsva_so_tree_tables()
        self.so_tree_tab_offset = len(vals)
        vals += fl
        self.gen_add_code_line("// tree topology for the second-order kernels, %d ints per lane: parent, tree level, subtree size, number of children, children" % K)
        self.gen_add_code_line("__device__ const int grid_so_tree_topology[%d] = {%s};" % (len(it), ", ".join(str(int(x)) for x in it)))
    if self.gen_idsva_so_mode() is not None and self.tuning["so_mapping"] == "balanced":
        it = self.gen_idsva_so_items_table()
        self.gen_add_code_line("// work items (c, m) of the second-order main loops, [lane][slot] (algorithms/_idsva_so.py: gen_idsva_so_items)")
        self.gen_add_code_line("__device__ const int grid_so_items[%d] = {%s};" % (len(it), ", ".join(str(int(x)) for x in it)))
    if self.gen_idsva_so_mode() is not None and self.gen_fdsva_so_components() is not None:
        flat, _ = self.gen_fdsva_so_components()
        self.gen_add_code_line("// base-rooted component of every joint: first joint, size (algorithms/_fdsva_so.py: the contraction of fdsva_so is block diagonal over the components)")
        self.gen_add_code_line("__device__ const int grid_so_component[%d] = {%s};" % (len(flat), ", ".join(str(int(x)) for x in flat)))
    if self.gen_idsva_so_mode() is not None and self.gen_idsva_so_blocks():
        Lb = self.gen_idsva_so_blocks_layout()
        self.gen_add_code_line("// block staging of the second-order record: first joint, size and block base of every joint's base-rooted component (algorithms/_idsva_so.py: gen_idsva_so_blocks_layout)")
        self.gen_add_code_line("__device__ const int grid_so_blocks[%d] = {%s};" % (len(Lb["JOINTS"]), ", ".join(str(int(x)) for x in Lb["JOINTS"])))
    if self.gen_idsva_so_mode() is not None and self.gen_idsva_so_packed():
        L = self.gen_idsva_so_compact_layout()
        self.gen_add_code_line("// slot of every element of the dense second-order record in the compact staging record (algorithms/_idsva_so.py: gen_idsva_so_compact_layout)")
        self.gen_add_code_line("__device__ __attribute__((aligned(16))) const unsigned short grid_so_expand[%d] = {%s};" % (len(L["TABLE"]), ", ".join(str(int(x)) for x in L["TABLE"])))
    for ctype, sfx in (("float", "f"), ("double", "")):
        self.gen_add_code_line("__device__ const %s grid_model_constants_%s[%d] = {" % (ctype, ctype, len(vals)), True)
        for k in range(0, len(vals), 6):
            self.gen_add_code_line(", ".join(repr(float(v)) + sfx for v in vals[k:k + 6]) + ("," if k + 6 < len(vals) else ""))
        self.indent_level -= 1
        self.gen_add_code_line("};")
    self.gen_add_code_line("__device__ __forceinline__ const float *grid_model_constants(const float *) { return grid_model_constants_float; }")
    self.gen_add_code_line("__device__ __forceinline__ const double *grid_model_constants(const double *) { return grid_model_constants_double; }")
    self.gen_add_code_line("")


def gen_init_XImats(self, include_base_inertia=False, include_homogenous_transforms=False):
    m = self.model
    n = m.n
    self.gen_add_func_doc("Initializes the model constant table in GPU memory",
                          ["Memory order is Xtree[0...N] (compact: E row-major 3x3, then B row-major 3x3), I[0...N] (6x6 column-major)"],
                          [], "A pointer to the XI memory in the GPU")
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__host__")
    self.gen_add_code_line("T* init_XImats() {", True)
    size = self.gen_get_XI_size()
    self.gen_add_code_line("T *h_XImats = (T *)malloc(" + str(size) + "*sizeof(T));")
    for i in range(n):
        XT = m.X_tree[i]
        vals = [XT[r, c] for r in range(3) for c in range(3)] + [XT[3 + r, c] for r in range(3) for c in range(3)]
        self.gen_add_code_line("// Xtree[" + str(i) + "]")
        for k, v in enumerate(vals):
            self.gen_add_code_line("h_XImats[" + str(18 * i + k) + "] = static_cast<T>(" + repr(float(v)) + ");")
    for i in range(n):
        self.gen_add_code_line("// I[" + str(i) + "]")
        for col in range(6):
            for row in range(6):
                self.gen_add_code_line("h_XImats[" + str(18 * n + 36 * i + 6 * col + row) + "] = static_cast<T>(" + repr(float(m.I[i][row, col])) + ");")
    self.gen_add_code_line("T *d_XImats; gpuErrchk(hipMalloc((void**)&d_XImats," + str(size) + "*sizeof(T)));")
    self.gen_add_code_line("gpuErrchk(hipMemcpy(d_XImats,h_XImats," + str(size) + "*sizeof(T),hipMemcpyHostToDevice));")
    self.gen_add_code_line("free(h_XImats);")
    self.gen_add_code_line("return d_XImats;")
    self.gen_add_end_function()


def gen_init_topology_helpers(self):
    m = self.model
    n = m.n
    self.gen_add_func_doc("Initializes the topology table in GPU memory",
                          ["Memory order is parent_id[0...N], S_index[0...N]; the generated kernels bake the topology into the",
                           "instruction stream and do not read this table - it is kept for downstream code that walks the tree"],
                          [], "A pointer to the topology_helpers memory in the GPU")
    self.gen_add_code_line("__host__")
    self.gen_add_code_line("inline int *init_topology_helpers(){", True)
    vals = [str(p) for p in m.parent] + [str(s) for s in m.S_index]
    self.gen_add_code_line("int h_topology_helpers[] = {" + ",".join(vals) + "};")
    self.gen_add_code_line("int *d_topology_helpers; gpuErrchk(hipMalloc((void**)&d_topology_helpers," + str(2 * n) + "*sizeof(int)));")
    self.gen_add_code_line("gpuErrchk(hipMemcpy(d_topology_helpers,h_topology_helpers," + str(2 * n) + "*sizeof(int),hipMemcpyHostToDevice));")
    self.gen_add_code_line("return d_topology_helpers;")
    self.gen_add_end_function()


def gen_init_robotModel(self):
    self.gen_add_func_doc("Initializes the robotModel helpers in GPU memory", [], [], "A pointer to the robotModel struct in GPU memory")
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__host__")
    self.gen_add_code_line("robotModel<T>* init_robotModel() {", True)
    self.gen_add_code_lines(["robotModel<T> h_robotModel;",
                             "h_robotModel.d_XImats = init_XImats<T>();",
                             "h_robotModel.d_topology_helpers = init_topology_helpers();",
                             "robotModel<T> *d_robotModel; gpuErrchk(hipMalloc((void**)&d_robotModel,sizeof(robotModel<T>)));",
                             "gpuErrchk(hipMemcpy(d_robotModel,&h_robotModel,sizeof(robotModel<T>),hipMemcpyHostToDevice));",
                             "return d_robotModel;"])
    self.gen_add_end_function()


def gen_load_update_XImats_helpers_function_call(self, use_thread_group=False, updated_var_names=None):
    var = dict(s_X_name="s_X", s_q_name="s_q")
    if updated_var_names is not None:
        var.update(updated_var_names)
    self.gen_add_code_line("load_update_XImats_helpers<T>(" + var["s_X_name"] + ", " + var["s_q_name"] + ", d_robotModel, lane);")
    self.gen_add_sync(use_thread_group)


def gen_load_update_XImats_helpers(self, use_thread_group=False):
    """Lane j evaluates X_j(q_j) = X_J(q_j) * Xtree_j into this solve's LDS slice (18 floats)."""
    m = self.model
    n = m.n
    self.gen_add_code_lines([
        "// sine and cosine of a joint angle.  fp32: three-constant Cody-Waite reduction by pi/2 (exact products inside the FMAs) and the Cephes minimax",
        "// polynomials on [-pi/4, pi/4]: max error 1.2 ulp(1) for |x| <= 1e5 (checked against double precision on 10^7 points), 29 instructions and no",
        "// branch in the common case - the math library's sincosf is 120 instructions and three branches.  Larger arguments (and non-finite ones) take the library path.",
        "__device__ __forceinline__ void grid_sincos(const float x, float *s, float *c) {",
        "    if (%s) { sincosf(x, s, c); return; }" % ("!(fabsf(x) <= 65536.0f)" if self.tuning["fast_sincos"] else "true"),
        "    const float k = rintf(x*0.6366197466850281f);",
        "    float r = fmaf(k, -1.5707963705062866f, x);        // pi/2 = hi + mid + lo",
        "    r = fmaf(k, 4.371138828673793e-08f, r);",
        "    r = fmaf(k, 1.7151245100058819e-15f, r);",
        "    const float r2 = r*r;",
        "    float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);",
        "    sp = fmaf(sp, r2, -1.6666654611e-1f);",
        "    const float sn = fmaf(r*r2, sp, r);",
        "    float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);",
        "    cp = fmaf(cp, r2, 4.166664568298827e-2f);",
        "    cp = fmaf(cp, r2, -0.5f);",
        "    const float cs = fmaf(cp, r2, 1.0f);",
        "    const int q = static_cast<int>(k);",
        "    const float a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;",
        "    *s = (q & 2) ? -a : a;",
        "    *c = ((q + 1) & 2) ? -b : b;",
        "}",
        "__device__ __forceinline__ void grid_sincos(const double x, double *s, double *c) { sincos(x, s, c); }", ""])
    self.gen_add_func_doc("Updates the joint transforms X(q) of one solve in LDS",
                          ["lane j of the solve's lane group handles joint j; only the rows of E and B that the joint motion mixes are",
                           "recomputed (12 of 18 numbers per joint), the remaining row is copied from the constant table"],
                          ["s_X is this solve's compact transform storage (GRID_X_STRIDE floats per joint)",
                           "s_q is the vector of joint positions in LDS",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU",
                           "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void load_update_XImats_helpers(T *s_X, const T *s_q, const robotModel<T> *d_robotModel, const int lane) {", True)
    self.gen_add_code_line("if (lane < " + str(n) + ") {", True)
    self.gen_add_code_lines(["const T *XT = &grid_model_constants(static_cast<const T *>(nullptr))[18*lane]; (void)d_robotModel;",
                             "T *Xo = &s_X[GRID_X_STRIDE*lane];",
                             "const T q = s_q[lane];"])
    types = sorted(set(m.S_index))
    if len(types) > 1 and any(s_ < 3 for s_ in types):  # one sine/cosine for all revolute lanes instead of one per axis block (the blocks run one after the other)
        self.gen_add_code_line("T sn, cs; grid_sincos(q, &sn, &cs);")
    first = True
    for s in types:
        ids = [j for j in range(n) if m.S_index[j] == s]
        cond = "" if len(types) == 1 else (("if " if first else "else if ") + self.gen_lane_mask_test(ids) + " ")
        first = False
        a = s % 3
        r1, r2 = (a + 1) % 3, (a + 2) % 3
        self.gen_add_code_line(cond + "{ // " + ("revolute" if s < 3 else "prismatic") + " about " + "xyz"[a] + ": joints " + str(ids), True)
        if s < 3:
            if len(types) == 1:
                self.gen_add_code_line("T sn, cs; grid_sincos(q, &sn, &cs);")
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int c = 0; c < 3; c++) {", True)
            for blk in (0, 9):
                self.gen_add_code_line("Xo[%d+c] = cs*XT[%d+c] + sn*XT[%d+c];" % (blk + 3 * r1, blk + 3 * r1, blk + 3 * r2))
                self.gen_add_code_line("Xo[%d+c] = cs*XT[%d+c] - sn*XT[%d+c];" % (blk + 3 * r2, blk + 3 * r2, blk + 3 * r1))
                self.gen_add_code_line("Xo[%d+c] = XT[%d+c];" % (blk + 3 * a, blk + 3 * a))
            self.gen_add_end_control_flow()
        else:
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int c = 0; c < 9; c++) { Xo[c] = XT[c]; }")
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int c = 0; c < 3; c++) {", True)
            self.gen_add_code_line("Xo[%d+c] = XT[%d+c] + q*XT[%d+c];" % (9 + 3 * r1, 9 + 3 * r1, 3 * r2))
            self.gen_add_code_line("Xo[%d+c] = XT[%d+c] - q*XT[%d+c];" % (9 + 3 * r2, 9 + 3 * r2, 3 * r1))
            self.gen_add_code_line("Xo[%d+c] = XT[%d+c];" % (9 + 3 * a, 9 + 3 * a))
            self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
    self.gen_add_end_control_flow()
    self.gen_add_end_function()


def gen_topology_sparsity_helpers_python(self, INIT_MODE=False):
    """Column bookkeeping of the reference's sparsity-compressed derivative layout, same return values as the reference
    (helpers/_topology_helpers.py:515-542).  Our kernels do not need it (joint-ordered column slots + structural zeros);
    it is kept as a topology query for code written against the reference's Code Generation API."""
    m = self.model
    n = m.n
    num_anc = [len(a) for a in m.ancestors]
    num_sub = [len(s_) for s_ in m.subtree]
    run_anc = [sum(num_anc[:j]) for j in range(n + 1)]
    run_sub = [sum(num_sub[:j]) for j in range(n)]
    if INIT_MODE:
        return [str(x) for x in num_anc], [str(x) for x in num_sub], [str(x) for x in run_anc], [str(x) for x in run_sub]
    dva_cols_per_partial = sum(num_anc) + n
    df_cols_per_partial = sum(num_anc) + sum(num_sub)
    dva_cols_per_jid = [x + 1 for x in num_anc]
    df_cols_per_jid = [num_anc[j] + num_sub[j] for j in range(n)]
    running_sum_dva_cols_per_jid = [run_anc[j] + j for j in range(n + 1)]
    running_sum_df_cols_per_jid = [run_anc[j] + run_sub[j] for j in range(n)]
    return dva_cols_per_partial, dva_cols_per_jid, running_sum_dva_cols_per_jid, df_cols_per_partial, df_cols_per_jid, running_sum_df_cols_per_jid, list(num_anc)


def gen_topology_helpers_pointers_for_cpp(self, inds=None, updated_var_names=None, NO_GRAD_FLAG=True, OFFSET=True):
    """C++ expressions for the parent id and motion-subspace index of joint `jid` (reference helpers/_topology_helpers.py:592-681).
    For a single joint the values are literals; otherwise they index the [parent | S_index] table of init_topology_helpers().
    Only the NO_GRAD form is offered: the compressed-column offsets of the reference layout have no counterpart here."""
    var = dict(jid_name="jid", s_topology_helpers_name="s_topology_helpers")
    if updated_var_names is not None:
        var.update(updated_var_names)
    m = self.model
    n = m.n
    if inds is None:
        inds = list(range(n))
    if not NO_GRAD_FLAG:
        raise NotImplementedError("gradient column offsets belong to the reference's compressed layout; see gen_gradient_slots()")
    if len(inds) == 1:
        return str(m.parent[inds[0]]), str(m.S_index[inds[0]])
    if m.is_serial_chain():
        parent_ind = "(" + var["jid_name"] + "-1)"
    else:
        parent_ind = var["s_topology_helpers_name"] + "[" + var["jid_name"] + "]"
    if len(set(m.S_index[i] for i in inds)) == 1:
        S_ind = str(m.S_index[inds[0]])
    else:
        S_ind = var["s_topology_helpers_name"] + "[" + str(n) + " + " + var["jid_name"] + "]"
    return parent_ind, S_ind
