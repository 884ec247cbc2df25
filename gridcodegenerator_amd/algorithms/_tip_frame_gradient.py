"""Forward-dynamics gradient for serial revolute chains in ONE common frame (the tip link's), emitter for the HIP/CDNA4 backend.

Same function as the column walk of algorithms/_inverse_dynamics_gradient.py (reference algorithms/_inverse_dynamics_gradient.py:27-775
and _forward_dynamics_gradient.py:7-62; oracle /root/reference/_test.py:229-520), different algorithm: instead of pushing 2n derivative
columns through every link transform (O(n d) 6x6 transforms), all link quantities are expressed once in a single frame F and the
partial derivatives of RNEA are assembled from per-joint vectors by dot products (the inverse-dynamics derivative identities of
Singh, Russell & Wensing, "Efficient Analytical Derivatives of Rigid-Body Dynamics using Spatial Vector Algebra", RA-L 2022):

    with  Pd_j  = v_j x S_j,  Pdd_j = a_j x S_j + v_j x Pd_j  (motion cross products, all vectors in F coordinates)
          I^C, B^C, f^C = sums over the links j..n-1 of  I_k,  B(I_k, v_k) = (v x*) I - I (v x) + (I v) xbar*,  f_k = I_k a_k + v_k x* I_k v_k
          t1 = I^C S_j   t2 = B^C S_j + 2 I^C Pd_j   t3 = B^C Pd_j + I^C Pdd_j + S_j x* f^C   t4 = (B^C)^T S_j
    for k <= j:   d c_k/d q_j = S_k . t3_j                 d c_k/d qd_j = S_k . t2_j
    for k >  j:   d c_k/d q_j = t1_k . Pdd_j + t4_k . Pd_j   d c_k/d qd_j = 2 t1_k . Pd_j + t4_k . S_j
    B(I, v) of a rigid body is [[Sym - n~, 0], [-2 l~, 0]] with [n; l] = I v and Sym = w~ Ibar + (w~ Ibar)^T - h u^T - u h^T + 2 (u.h) 1,
    i.e. 12 numbers, and I^C stays a 10-parameter rigid-body inertia.

Why the TIP frame: every d c_k/d u_j is later multiplied by M^-1, whose entries are ~1e3 for the light wrist links and ~1 for the
base links.  In the world frame the wrist rows are small differences of large moments about a far-away origin and fp32 loses
4 digits (measured: max error 1.8e-4 of max|df/du| on the 7-DoF arm); with the origin at the tip the same rows are computed
about a nearby origin and the error is that of the link-local recursion (3e-6).  F is the inertial frame that coincides with
the tip link at this instant, so all identities hold unchanged.

Lane mapping (chain: lane j <-> joint j, parent = lane j-1): the frame chain tip -> base is computed wave-uniformly from the rotation
blocks of X(q) and every lane keeps its own (R_j, p_j); prefix sums over the ancestors (v, a) and suffix sums over the subtree
(I^C, B^C, f^C) are log-step DPP scans inside the lane group; the only LDS hand-offs are one 16-float record per joint
[S | t1 | t4 | tau - c] that every lane reads to assemble its two columns, and the n x n joint-space inertia.

The joint-space inertia needs nothing new: M[k][j] = S_k . (I^C_j S_j) = S_k . t1_j for k <= j.  It is never inverted: every lane factors
it in registers (U D U^T, wave-uniform) and solves for tau - c (-> qdd) and for its own two columns of dc/du.  The stand-alone kernels
of such robots (inverse dynamics, its gradient, forward dynamics, M^-1) are subsets of the same code (gen_tip_frame_components).

Forests: a fixed-base robot whose limbs are separate chains hanging off the base (a quadruped's legs) is several independent solves of
this kind side by side in one lane group - every limb has its own tip frame, its own scans (segments of the lane group), its own
L x L inertia block; limbs must have the same length and joint-axis sequence so that one instruction stream serves them all.

Scope: serial chains of revolute joints (the reference's own iiwa case) and forests of equal such chains.  Robots with prismatic joints stay on the column walk: the
reference's oracle differs from the true derivative for non-root prismatic joints (checked by finite differences of its own RNEA),
and parity with the reference is the contract.
"""
import numpy as np


def gen_tip_frame_link_constants(self):
    """Per-lane table rows appended to grid_model_constants: [Ic_xx Ic_xy Ic_xz Ic_yy Ic_yz Ic_zz | c_x c_y c_z | m | damping | S axis];
    rows of lanes that hold no joint are zero (all suffix-summed quantities are linear in the inertia, so they contribute nothing)."""
    m = self.model
    rows = []
    for j in range(self.lanes_per_solve):
        if j >= m.n:
            rows += [0.0] * 12
            continue
        I = m.I[j]
        mass = I[3, 3]
        H = I[:3, 3:]
        h = np.array([H[2, 1], H[0, 2], H[1, 0]])
        c = h / mass
        Ic = I[:3, :3] - mass * (float(c @ c) * np.eye(3) - np.outer(c, c))
        rows += [Ic[0, 0], Ic[0, 1], Ic[0, 2], Ic[1, 1], Ic[1, 2], Ic[2, 2], c[0], c[1], c[2], mass, float(m.damping[j]), float(m.S_index[j])]
    if self.tip_nseg > 1:  # forests: the chain steps read the joint offsets from a table (they differ between limbs), 4 values per joint
        for j in range(m.n):
            r = self.gen_tip_frame_joint_offset(j)
            rows += [r[0], r[1], r[2], 0.0]
    return rows


def gen_tip_frame_joint_offset(self, j):
    """Origin of frame j in the coordinates of its parent frame (constant for a revolute joint): X_tree = [[E,0],[-E r~,E]] -> r."""
    XT = self.model.X_tree[j]
    E, B = XT[:3, :3], XT[3:, :3]
    rx = -E.T @ B
    r = np.array([rx[2, 1], rx[0, 2], rx[1, 0]])
    return np.where(np.abs(r) < 1e-12, 0.0, r)


_TIP_LIBRARY = r"""
// ---------------------------------------------------------------------------------------------------------------------
// tip-frame gradient path (serial revolute chains): cross-lane moves, scans, 10-parameter inertias
// ---------------------------------------------------------------------------------------------------------------------
// value held CTRL-encoded lanes away inside the 16-lane DPP row (0 when that lane is outside the row); no LDS traffic
template <int CTRL> __device__ __forceinline__ float grid_dpp_mov(const float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double grid_dpp_mov(const double x) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned int lo = static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<unsigned int>(b)), CTRL, 0xf, 0xf, true));
    const unsigned int hi = static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<unsigned int>(b >> 32)), CTRL, 0xf, 0xf, true));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}
template <int K, typename T> __device__ __forceinline__ T grid_lane_below(const T x) { return grid_dpp_mov<0x110 + K>(x); }  // row_shr:K : from lane - K
template <int K, typename T> __device__ __forceinline__ T grid_lane_above(const T x) { return grid_dpp_mov<0x100 + K>(x); }  // row_shl:K : from lane + K

// y = a x b for motion vectors ([w;v]): [wa x wb ; wa x vb + va x wb]
template <typename T>
__device__ __forceinline__ void grid_mxm(T (&y)[6], const T (&a)[6], const T (&b)[6]) {
    y[0] = a[1]*b[2] - a[2]*b[1];
    y[1] = a[2]*b[0] - a[0]*b[2];
    y[2] = a[0]*b[1] - a[1]*b[0];
    y[3] = a[1]*b[5] - a[2]*b[4] + a[4]*b[2] - a[5]*b[1];
    y[4] = a[2]*b[3] - a[0]*b[5] + a[5]*b[0] - a[3]*b[2];
    y[5] = a[0]*b[4] - a[1]*b[3] + a[3]*b[1] - a[4]*b[0];
}
template <typename T>
__device__ __forceinline__ void grid_mxm_peq(T (&y)[6], const T (&a)[6], const T (&b)[6]) {
    y[0] += a[1]*b[2] - a[2]*b[1];
    y[1] += a[2]*b[0] - a[0]*b[2];
    y[2] += a[0]*b[1] - a[1]*b[0];
    y[3] += a[1]*b[5] - a[2]*b[4] + a[4]*b[2] - a[5]*b[1];
    y[4] += a[2]*b[3] - a[0]*b[5] + a[5]*b[0] - a[3]*b[2];
    y[5] += a[0]*b[4] - a[1]*b[3] + a[3]*b[1] - a[4]*b[0];
}

// rigid-body inertia about the origin of the working frame, I = [Ixx Ixy Ixz Iyy Iyz Izz | hx hy hz | m]:  y (+)= I x = [Ibar w + h x u ; m u - h x w]
template <typename T>
__device__ __forceinline__ void grid_rbi_mul(T (&y)[6], const T (&I)[10], const T (&x)[6]) {
    y[0] = I[0]*x[0] + I[1]*x[1] + I[2]*x[2] + I[7]*x[5] - I[8]*x[4];
    y[1] = I[1]*x[0] + I[3]*x[1] + I[4]*x[2] + I[8]*x[3] - I[6]*x[5];
    y[2] = I[2]*x[0] + I[4]*x[1] + I[5]*x[2] + I[6]*x[4] - I[7]*x[3];
    y[3] = I[9]*x[3] - I[7]*x[2] + I[8]*x[1];
    y[4] = I[9]*x[4] - I[8]*x[0] + I[6]*x[2];
    y[5] = I[9]*x[5] - I[6]*x[1] + I[7]*x[0];
}
template <typename T>
__device__ __forceinline__ void grid_rbi_mul_peq(T (&y)[6], const T (&I)[10], const T (&x)[6], const T alpha) {
    y[0] += alpha*(I[0]*x[0] + I[1]*x[1] + I[2]*x[2] + I[7]*x[5] - I[8]*x[4]);
    y[1] += alpha*(I[1]*x[0] + I[3]*x[1] + I[4]*x[2] + I[8]*x[3] - I[6]*x[5]);
    y[2] += alpha*(I[2]*x[0] + I[4]*x[1] + I[5]*x[2] + I[6]*x[4] - I[7]*x[3]);
    y[3] += alpha*(I[9]*x[3] - I[7]*x[2] + I[8]*x[1]);
    y[4] += alpha*(I[9]*x[4] - I[8]*x[0] + I[6]*x[2]);
    y[5] += alpha*(I[9]*x[5] - I[6]*x[1] + I[7]*x[0]);
}

// body-level Coriolis matrix B = [[Sym - n~, 0], [-2 l~, 0]] stored as [Sxx Sxy Sxz Syy Syz Szz | nx ny nz | lx ly lz]
// y = B x : only the angular part of x matters
template <typename T>
__device__ __forceinline__ void grid_bmul(T (&y)[6], const T (&B)[12], const T (&x)[6]) {
    y[0] = B[0]*x[0] + B[1]*x[1] + B[2]*x[2] - (B[7]*x[2] - B[8]*x[1]);
    y[1] = B[1]*x[0] + B[3]*x[1] + B[4]*x[2] - (B[8]*x[0] - B[6]*x[2]);
    y[2] = B[2]*x[0] + B[4]*x[1] + B[5]*x[2] - (B[6]*x[1] - B[7]*x[0]);
    y[3] = static_cast<T>(-2)*(B[10]*x[2] - B[11]*x[1]);
    y[4] = static_cast<T>(-2)*(B[11]*x[0] - B[9]*x[2]);
    y[5] = static_cast<T>(-2)*(B[9]*x[1] - B[10]*x[0]);
}
// y = top half of B^T x (the bottom half is zero): Sym xw + n x xw + 2 l x xu
template <typename T>
__device__ __forceinline__ void grid_btmul(T (&y)[3], const T (&B)[12], const T (&x)[6]) {
    y[0] = B[0]*x[0] + B[1]*x[1] + B[2]*x[2] + (B[7]*x[2] - B[8]*x[1]) + static_cast<T>(2)*(B[10]*x[5] - B[11]*x[4]);
    y[1] = B[1]*x[0] + B[3]*x[1] + B[4]*x[2] + (B[8]*x[0] - B[6]*x[2]) + static_cast<T>(2)*(B[11]*x[3] - B[9]*x[5]);
    y[2] = B[2]*x[0] + B[4]*x[1] + B[5]*x[2] + (B[6]*x[1] - B[7]*x[0]) + static_cast<T>(2)*(B[9]*x[4] - B[10]*x[3]);
}
"""


def gen_tip_frame_library(self):
    """Generic device helpers of the tip-frame path plus the lane-group scans (their step count depends on GRID_LANES_PER_SOLVE)."""
    for line in _TIP_LIBRARY.strip("\n").split("\n"):
        self.gen_add_code_line(line)
    L = self.tip_L
    steps = [k for k in (1, 2, 4, 8) if k < L]
    ns = max(1, len(steps))
    # interleaved 8-lane groups holding ONE chain: the lanes of a solve are every other lane of a 16-lane DPP row, joint j's neighbours are 2 lanes away and
    # the row ends where the chain ends - a row shift with bound_ctrl reads 0 beyond it, so the steps need no masks and a lane never sees the other solve's
    # values (a masked multiply would turn that solve's NaN / Inf into a NaN here: 0 * NaN).  Lanes without a joint hold exact zeros (zero link constants).
    stride = 2 if getattr(self, "lane_interleave", False) else 1
    unmasked = stride == 2 and self.tip_nseg == 1 and not getattr(self, "branch_frame", False)
    self.gen_add_code_line("")
    self.gen_add_code_line("// inclusive sums over the lanes of one chain (lane j <-> joint j; pos = position of the joint in its chain of %d): prefix = over joint j" % L)
    self.gen_add_code_line("// and its ancestors, suffix = over joint j and its descendants.  Log-step DPP scans" + ("; the lanes of a solve are every other lane of a 16-lane row, which ends where the chain ends: no masks" if unmasked else "; mk[s] is 1 where the partner lane of step s is in the same chain."))
    self.gen_add_code_line("#define GRID_SCAN_STEPS %d" % ns)
    for name, fn, cmp_ in (("prefix", "grid_lane_below", "pos >= %d"), ("suffix", "grid_lane_above", "pos + %d < %d")):
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__device__ __forceinline__ void grid_%s_masks(T (&mk)[GRID_SCAN_STEPS], const int pos) {" % name, True)
        if not steps:
            self.gen_add_code_line("mk[0] = static_cast<T>(0); (void)pos;")
        for s_, k in enumerate(steps):
            cond = (cmp_ % k) if name == "prefix" else (cmp_ % (k, L))
            self.gen_add_code_line("mk[%d] = (%s) ? static_cast<T>(1) : static_cast<T>(0);" % (s_, cond))
        self.gen_add_end_function()
        self.gen_add_code_line("template <int N, typename T>")
        self.gen_add_code_line("__device__ __forceinline__ void grid_%s_sum(T (&x)[N], const T (&mk)[GRID_SCAN_STEPS]) {" % name, True)
        if not steps or unmasked:
            self.gen_add_code_line("(void)x; (void)mk;")
        for s_, k in enumerate(steps):
            self.gen_add_code_line("#pragma unroll")
            if unmasked:
                self.gen_add_code_line("for (int r = 0; r < N; r++) { x[r] += %s<%d>(x[r]); }" % (fn, k * stride))
            else:
                self.gen_add_code_line("for (int r = 0; r < N; r++) { x[r] += mk[%d]*%s<%d>(x[r]); }" % (s_, fn, k * stride))
        self.gen_add_end_function()
    if self.tuning["dpp_asm"]:
        # fp32 device code: one v_fmac_f32_dpp per value and step instead of the v_mov_b32_dpp + v_fma pair the compiler emits for the
        # builtin (it does not fold the DPP move into the FMA).  The hazard recognizer does not look inside inline asm, so every block
        # opens with the 2 wait states a DPP read needs after a VALU write of the same register.
        self.gen_add_code_line("#if defined(__HIP_DEVICE_COMPILE__)")
        for name, ctrl in (("prefix", "row_shr"), ("suffix", "row_shl")):
            for N in (6, 10, 12):
                self.gen_add_code_line("__device__ __forceinline__ void grid_%s_sum(float (&x)[%d], const float (&mk)[GRID_SCAN_STEPS]) {" % (name, N), True)
                if not steps:
                    self.gen_add_code_line("(void)x; (void)mk;")
                if unmasked:
                    self.gen_add_code_line("(void)mk;")
                for s_, k in enumerate(steps):
                    for lo in range(0, N, 6):
                        cnt = min(6, N - lo)
                        outs = ", ".join('"+v"(x[%d])' % (lo + r) for r in range(cnt))
                        if unmasked and self.tuning["scan_form"] == "fmac":  # x += 1 * (the value 2k lanes away, 0 beyond the row)
                            body = "s_nop 1" + "".join("\\n\\tv_fmac_f32_dpp %%%d, %%%d, %%%d %s:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % (r, r, cnt, ctrl, k * stride) for r in range(cnt))
                            self.gen_add_code_line('{ const float one = 1.0f; asm("%s" : %s : "v"(one)); }' % (body, outs))
                        elif unmasked:  # x += (the value 2k lanes away, 0 beyond the row): one v_add_f32_dpp
                            body = "s_nop 1" + "".join("\\n\\tv_add_f32_dpp %%%d, %%%d, %%%d %s:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % (r, r, r, ctrl, k * stride) for r in range(cnt))
                            self.gen_add_code_line('asm("%s" : %s);' % (body, outs))
                        else:
                            body = "s_nop 1" + "".join("\\n\\tv_fmac_f32_dpp %%%d, %%%d, %%%d %s:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % (r, r, cnt, ctrl, k * stride) for r in range(cnt))
                            self.gen_add_code_line('asm("%s" : %s : "v"(mk[%d]));' % (body, outs, s_))
                self.gen_add_end_function()
        self.gen_add_code_line("#endif")
    self.gen_add_code_line("")


def _probe(self, stage, *arrays):
    """Accuracy diagnosis only (tuning round_probe): round the named register arrays ("name:len") to fp32 at the end of a stage."""
    if stage not in tuple(self.tuning.get("round_probe", ())):
        return
    for a in arrays:
        nm, ln = a.split(":")
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < %s; r++) { %s[r] = static_cast<T>(static_cast<float>(%s[r])); } // round_probe %s" % (ln, nm, nm, stage))


def _chain_step(self, i, s_F=None, with_gravity=True, base_family=False):
    """Frame-chain step of the joint at position i of every chain: hand (R, p) to the lane that owns that joint, then move the running
    frame to the parent (or, for the root, read off the world's gravity direction).  The hand-off is a register select on every lane
    (s_F None) or one LDS record per joint written by lane 0 and read back by its owner after the chain (single chains only)."""
    m = self.model
    L, nseg = self.tip_L, self.tip_nseg
    E_nz = m.X_pattern[i][0].copy()
    for sgi in range(1, nseg):
        E_nz |= m.X_pattern[sgi * L + i][0]
    C = lambda x: "static_cast<T>(" + repr(float(x)) + ")"
    J = str(i) if nseg == 1 else "(base + %d)" % i
    self.gen_add_code_line("{ // tip-frame chain, position %d: the lane of that joint keeps (R, p) of its frame; then on to the frame of its parent" % i, True)
    if base_family and self.tip_jB is not None and i == self.tip_jB:
        self.gen_add_code_line("pB[0] = pc[0]; pB[1] = pc[1]; pB[2] = pc[2]; // origin of the base family: the frame origin of the joint at position %d" % i)
    if s_F is None:
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 9; r++) { myR[r] = (pos == %d) ? Rc[r] : myR[r]; }" % i)
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 3; r++) { myp[r] = (pos == %d) ? pc[r] : myp[r]; }" % i)
    elif i < L - 1:  # (the tip's own frame is the identity: its lane keeps the initial values)
        self.gen_add_code_line("if (lane == 0) {", True)
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 9; r++) { %s[%d + r] = Rc[r]; }" % (s_F, 16 * i))
        self.gen_add_code_line("%s[%d] = pc[0]; %s[%d] = pc[1]; %s[%d] = pc[2];" % (s_F, 16 * i + 9, s_F, 16 * i + 10, s_F, 16 * i + 11))
        self.gen_add_end_control_flow()
    self.gen_add_code_line("const T *Ei = &s_X[GRID_X_STRIDE*%s]; // E(q) of that joint: parent -> child coordinates (row-major)" % J)
    self.gen_add_code_line("T Rn[9];")
    for rr in range(3):
        for cc in range(3):
            terms = ["Rc[%d]*Ei[%d]" % (3 * rr + k, 3 * k + cc) for k in range(3) if E_nz[k, cc]]
            self.gen_add_code_line("Rn[%d] = %s;" % (3 * rr + cc, " + ".join(terms) if terms else "static_cast<T>(0)"))
    if i > 0:
        if nseg == 1:
            r = self.gen_tip_frame_joint_offset(i)
            for rr in range(3):
                terms = ["Rn[%d]*%s" % (3 * rr + k, C(r[k])) for k in range(3) if r[k] != 0.0]
                if terms:
                    self.gen_add_code_line("pc[%d] -= %s;" % (rr, " + ".join(terms)))
        else:
            self.gen_add_code_line("const T *rj = &grid_model_constants(static_cast<const T *>(nullptr))[%d + 4*%s]; // origin of that joint's frame in its parent's coordinates" % (54 * m.n + 12 * self.lanes_per_solve, J))
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 3; r++) { pc[r] -= Rn[3*r]*rj[0] + Rn[3*r+1]*rj[1] + Rn[3*r+2]*rj[2]; }")
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 9; r++) { Rc[r] = Rn[r]; }")
    elif with_gravity:
        self.gen_add_code_line("gvec[0] = gravity*Rn[2]; gvec[1] = gravity*Rn[5]; gvec[2] = gravity*Rn[8]; // base acceleration (0,0,g) in F coordinates")
    else:
        self.gen_add_code_line("(void)Rn; gvec[0] = gvec[1] = gvec[2] = static_cast<T>(0);")
    self.gen_add_end_control_flow()


def _emit_chain_decls(self):
    self.gen_add_code_line("// running frame of the tip -> base chain (wave-uniform) and this lane's own frame: R maps link coordinates to F, p is the link origin in F")
    self.gen_add_code_line("T Rc[9] = {static_cast<T>(1), static_cast<T>(0), static_cast<T>(0), static_cast<T>(0), static_cast<T>(1), static_cast<T>(0), static_cast<T>(0), static_cast<T>(0), static_cast<T>(1)};")
    self.gen_add_code_line("T pc[3] = {static_cast<T>(0), static_cast<T>(0), static_cast<T>(0)};")
    self.gen_add_code_line("T myR[9], myp[3], gvec[3];")
    if self.tip_jB is not None:
        self.gen_add_code_line("T pB[3] = {static_cast<T>(0), static_cast<T>(0), static_cast<T>(0)}; // second reference point (see _emit_base_family)")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 9; r++) { myR[r] = Rc[r]; }")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 3; r++) { myp[r] = pc[r]; }")


def _emit_link_setup(self, kinematics=True, base_family=False):
    """Per lane: joint axis S and link inertia in F coordinates; with kinematics also the velocity prefix sum and Pd = S-dot."""
    m = self.model
    n = m.n
    same_axis = len(set(m.S_index)) == 1
    self.gen_add_code_line("//")
    self.gen_add_code_line("// tip frame, lane j <-> joint j: joint axis, link inertia, velocity, bias acceleration / force, Coriolis matrix")
    self.gen_add_code_line("//")
    self.gen_add_code_line("T mku[GRID_SCAN_STEPS], mkd[GRID_SCAN_STEPS]; grid_prefix_masks(mku, pos); grid_suffix_masks(mkd, pos);")
    if kinematics:
        self.gen_add_code_line("const T qd = (lane < %d) ? s_qd[lane] : static_cast<T>(0); // lanes without a joint must contribute exact zeros to the scans (0 * uninitialised LDS may be NaN)" % n)
    self.gen_add_code_line("T S[6];")
    if same_axis:
        a = m.S_index[0]
        self.gen_add_code_line("S[0] = myR[%d]; S[1] = myR[%d]; S[2] = myR[%d]; // rotation axis %s of the link frame" % (a, 3 + a, 6 + a, "xyz"[a]))
    else:
        self.gen_add_code_line("{ const int ax = static_cast<int>(Lc[11]);")
        self.gen_add_code_line("  #pragma unroll")
        self.gen_add_code_line("  for (int r = 0; r < 3; r++) { S[r] = (ax == 0) ? myR[3*r] : ((ax == 1) ? myR[3*r+1] : myR[3*r+2]); } }")
    self.gen_add_code_line("S[3] = myp[1]*S[2] - myp[2]*S[1]; S[4] = myp[2]*S[0] - myp[0]*S[2]; S[5] = myp[0]*S[1] - myp[1]*S[0];")
    _emit_link_inertia(self, base_family)
    if not kinematics:
        return
    self.gen_add_code_line("T v[6];")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 6; r++) { v[r] = S[r]*qd; }")
    self.gen_add_code_line("grid_prefix_sum(v, mku); // v_j = sum over the ancestors of S_k qd_k")
    self.gen_add_code_line("T Pd[6]; grid_mxm(Pd, v, S); // = S_j-dot")


def _emit_link_inertia(self, base_family=False, local_origin=False):
    """I[10]: the lane's link inertia about the origin of the working frame, from (myR, myp) and the link constants Lc
    (local_origin: about the origin of the link's own joint frame, in F's axes - the tree form of the second-order kernels)."""
    self.gen_add_code_line("T I[10]; // this link's inertia about the origin of " + ("its own joint frame (axes of F)" if local_origin else "F"))
    if base_family and self.tip_jB is not None:
        self.gen_add_code_line("T IB[10], SB[3]; // ... and about pB, and the linear part of S about pB (base family, see _emit_base_family)")
    self.gen_add_code_line("{", True)
    self.gen_add_code_line("T d[3], RI[9], rot[6];")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 3; r++) {", True)
    self.gen_add_code_line("d[r] = %smyR[3*r]*Lc[6] + myR[3*r+1]*Lc[7] + myR[3*r+2]*Lc[8]; // centre of mass" % ("" if local_origin else "myp[r] + "))
    self.gen_add_code_line("RI[3*r]   = myR[3*r]*Lc[0] + myR[3*r+1]*Lc[1] + myR[3*r+2]*Lc[2];")
    self.gen_add_code_line("RI[3*r+1] = myR[3*r]*Lc[1] + myR[3*r+1]*Lc[3] + myR[3*r+2]*Lc[4];")
    self.gen_add_code_line("RI[3*r+2] = myR[3*r]*Lc[2] + myR[3*r+1]*Lc[4] + myR[3*r+2]*Lc[5];")
    self.gen_add_end_control_flow()
    pairs = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))
    for k, (r_, c_) in enumerate(pairs):
        self.gen_add_code_line("rot[%d] = RI[%d]*myR[%d] + RI[%d]*myR[%d] + RI[%d]*myR[%d]; // (R Ic R^T)[%d][%d]" % (k, 3 * r_, 3 * c_, 3 * r_ + 1, 3 * c_ + 1, 3 * r_ + 2, 3 * c_ + 2, r_, c_))

    def parallel_axis(dst, dv):
        self.gen_add_code_line("{ const T md0 = Lc[9]*%s[0], md1 = Lc[9]*%s[1], md2 = Lc[9]*%s[2];" % (dv, dv, dv))
        for k, (r_, c_) in enumerate(pairs):
            if r_ == c_:
                o1, o2 = [x for x in range(3) if x != r_]
                self.gen_add_code_line("  %s[%d] = rot[%d] + md%d*%s[%d] + md%d*%s[%d];" % (dst, k, k, o1, dv, o1, o2, dv, o2))
            else:
                self.gen_add_code_line("  %s[%d] = rot[%d] - md%d*%s[%d];" % (dst, k, k, r_, dv, c_))
        self.gen_add_code_line("  %s[6] = md0; %s[7] = md1; %s[8] = md2; %s[9] = Lc[9]; }" % (dst, dst, dst, dst))

    parallel_axis("I", "d")
    if base_family and self.tip_jB is not None:
        self.gen_add_code_line("T dB[3] = {d[0] - pB[0], d[1] - pB[1], d[2] - pB[2]}; // centre of mass relative to pB")
        parallel_axis("IB", "dB")
        self.gen_add_code_line("{ const T e0 = myp[0] - pB[0], e1 = myp[1] - pB[1], e2 = myp[2] - pB[2];")
        self.gen_add_code_line("  SB[0] = e1*S[2] - e2*S[1]; SB[1] = e2*S[0] - e0*S[2]; SB[2] = e0*S[1] - e1*S[0]; }")
    self.gen_add_end_control_flow()


def _emit_bias(self, with_qdd):
    """a (qdd = 0 or with qdd), Iv, f, Coriolis matrix and the composite (suffix) sums; leaves IC[10], BC[12], fC[6], a[6]."""
    self.gen_add_code_line("T a[6];")
    self.gen_add_code_line("#pragma unroll")
    if with_qdd:
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { a[r] = Pd[r]*qd + S[r]*qdd; }")
    else:
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { a[r] = Pd[r]*qd; }")
    self.gen_add_code_line("grid_prefix_sum(a, mku);")
    self.gen_add_code_line("a[3] += gvec[0]; a[4] += gvec[1]; a[5] += gvec[2];")
    _emit_body_terms(self)
    _emit_inertia_composite(self)
    self.gen_add_code_line("grid_suffix_sum(BC, mkd); grid_suffix_sum(fC, mkd); // composites over the links j..n-1")


def _emit_body_terms(self):
    """Per link, from I (10), v, a: the force f = I a + v x* I v (into fC) and the body-level Coriolis matrix B(I, v) (into BC); declares IC, BC, fC."""
    self.gen_add_code_line("T IC[10], BC[12], fC[6];")
    self.gen_add_code_line("{", True)
    self.gen_add_code_line("T Iv[6]; grid_rbi_mul(Iv, I, v); // [n; l]: the link's momentum")
    self.gen_add_code_line("grid_rbi_mul(fC, I, a); grid_fxv_peq(fC, v, Iv);")
    self.gen_add_code_line("// Sym = A + A^T - h u^T - u h^T + 2 (u.h) 1  with  A = w~ Ibar")
    self.gen_add_code_line("const T A00 = v[1]*I[2] - v[2]*I[1], A01 = v[1]*I[4] - v[2]*I[3], A02 = v[1]*I[5] - v[2]*I[4];")
    self.gen_add_code_line("const T A10 = v[2]*I[0] - v[0]*I[2], A11 = v[2]*I[1] - v[0]*I[4], A12 = v[2]*I[2] - v[0]*I[5];")
    self.gen_add_code_line("const T A20 = v[0]*I[1] - v[1]*I[0], A21 = v[0]*I[3] - v[1]*I[1], A22 = v[0]*I[4] - v[1]*I[2];")
    self.gen_add_code_line("const T uh = static_cast<T>(2)*(v[3]*I[6] + v[4]*I[7] + v[5]*I[8]);")
    self.gen_add_code_line("BC[0] = static_cast<T>(2)*(A00 - I[6]*v[3]) + uh;")
    self.gen_add_code_line("BC[1] = A01 + A10 - I[6]*v[4] - I[7]*v[3];")
    self.gen_add_code_line("BC[2] = A02 + A20 - I[6]*v[5] - I[8]*v[3];")
    self.gen_add_code_line("BC[3] = static_cast<T>(2)*(A11 - I[7]*v[4]) + uh;")
    self.gen_add_code_line("BC[4] = A12 + A21 - I[7]*v[5] - I[8]*v[4];")
    self.gen_add_code_line("BC[5] = static_cast<T>(2)*(A22 - I[8]*v[5]) + uh;")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 6; r++) { BC[6 + r] = Iv[r]; }")
    self.gen_add_end_control_flow()


def _emit_inertia_composite(self):
    """IC = suffix sum of the link inertias I (about the origin of F).  The entries of the base links are ~m d^2 with d the distance to the tip
    (25 kg m^2 for the 7-DoF arm) while the joint-space inertia they end up in can be 1e-2: the three rounding steps of an fp32 log-step scan
    at that magnitude are the largest single contribution to the fp32 error of the path (tools/precision_probe.py).  composite_scan = f64
    adds the nine inertia entries in double precision (exact for these magnitudes) and rounds once."""
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 10; r++) { IC[r] = I[r]; }")
    if self.tuning["composite_scan"] == "f64":
        self.gen_add_code_line("{ // exact suffix sums of the 9 inertia entries (double accumulators), one rounding at the end; the mass stays in T")
        self.gen_add_code_line("  double ICd[9], mkdd[GRID_SCAN_STEPS];")
        self.gen_add_code_line("  #pragma unroll")
        self.gen_add_code_line("  for (int r = 0; r < 9; r++) { ICd[r] = static_cast<double>(I[r]); }")
        self.gen_add_code_line("  #pragma unroll")
        self.gen_add_code_line("  for (int r = 0; r < GRID_SCAN_STEPS; r++) { mkdd[r] = static_cast<double>(mkd[r]); }")
        self.gen_add_code_line("  grid_suffix_sum(ICd, mkdd);")
        self.gen_add_code_line("  #pragma unroll")
        self.gen_add_code_line("  for (int r = 0; r < 9; r++) { IC[r] = static_cast<T>(ICd[r]); }")
        self.gen_add_code_line("  T mC[1] = {I[9]}; grid_suffix_sum(mC, mkd); IC[9] = mC[0]; }")
    else:
        self.gen_add_code_line("grid_suffix_sum(IC, mkd);")


def _emit_base_family(self):
    """Second reference point for the joint-space inertia (fp32 accuracy; the T = double instantiation runs the same code).
    M[k][j] = S_k . (I^C_j S_j) is a moment about joint axis k.  About the tip - up to the whole arm's length away from the base joints - it is
    a small difference of terms of size m d^2 (25 kg m^2 for the 7-DoF arm against entries of 1e-2 .. 1): ~1e-6 relative error per entry in fp32,
    amplified up to 50x where two base axes are nearly parallel (shoulder + elbow lined up).  The columns j <= jB (the base half of the chain)
    therefore take their entries from composites about pB, the frame origin of joint jB, where those terms are an order of magnitude smaller;
    the columns of the wrist half keep the tip (their own neighbourhood).  Costs one more suffix scan of 9 values, one more I^C S product and 3
    more values per hand-off record.  Leaves t1m (the I^C S this lane's column of M is built from) and `fam` (1: base family)."""
    jB = self.tip_jB
    self.gen_add_code_line("// base family: composites of the link inertias about pB for the columns of M owned by the joints at positions <= %d" % jB)
    self.gen_add_code_line("const bool fam = (pos <= %d);" % jB)
    self.gen_add_code_line("T t1m[6];")
    self.gen_add_code_line("{", True)
    self.gen_add_code_line("T ICB[10], x[6], y[6];")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 10; r++) { ICB[r] = IB[r]; }")
    self.gen_add_code_line("grid_suffix_sum(ICB, mkd);")
    self.gen_add_code_line("x[0] = S[0]; x[1] = S[1]; x[2] = S[2]; x[3] = SB[0]; x[4] = SB[1]; x[5] = SB[2];")
    self.gen_add_code_line("grid_rbi_mul(y, ICB, x);")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 6; r++) { t1m[r] = fam ? y[r] : t1[r]; }")
    self.gen_add_end_control_flow()


def _own_rows(self, dst_fmt, val_fmt, zero="static_cast<T>(0)"):
    """Emit the n stores of one output column: rows of the lane's own chain take val_fmt % position, the others are structural zeros
    (dst_fmt % row gives the lvalue).  One chain: every row is an own row."""
    n, L, nseg = self.model.n, self.tip_L, self.tip_nseg
    for row in range(n):
        sr, kk = divmod(row, L)
        if nseg == 1:
            self.gen_add_code_line("%s = %s;" % (dst_fmt % row, val_fmt % kk))
        else:
            self.gen_add_code_line("%s = (seg == %d) ? %s : %s;" % (dst_fmt % row, sr, val_fmt % kk, zero))


def _emit_assembly(self, s_G="s_G", dst="s_df_du", minv="s_Minv"):
    """Pdd, t1..t4, the [S | t1 | t4] hand-off and this lane's two columns of dc/du, then -Minv*dc/du into the staging area."""
    m = self.model
    n = m.n
    ld = self.minv_ld
    self.gen_add_code_line("T Pdd[6]; grid_mxm(Pdd, a, S); grid_mxm_peq(Pdd, v, Pd);")
    self.gen_add_code_line("T t1[6], t2[6], t3[6], t4[3];")
    self.gen_add_code_line("grid_rbi_mul(t1, IC, S);")
    self.gen_add_code_line("grid_bmul(t2, BC, S); grid_rbi_mul_peq(t2, IC, Pd, static_cast<T>(2));")
    self.gen_add_code_line("grid_bmul(t3, BC, Pd); grid_rbi_mul_peq(t3, IC, Pdd, static_cast<T>(1)); grid_fxv_peq(t3, S, fC);")
    self.gen_add_code_line("grid_btmul(t4, BC, S);")
    self.gen_add_code_line("if (lane < %d) { // hand-off record of joint `lane`: [S | t1 | t4]" % n, True)
    self.gen_add_code_line("T *rec = &%s[16*lane];" % s_G)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 6; r++) { rec[r] = S[r]; rec[6 + r] = t1[r]; }")
    self.gen_add_code_line("rec[12] = t4[0]; rec[13] = t4[1]; rec[14] = t4[2]; rec[15] = static_cast<T>(0);")
    self.gen_add_end_control_flow()
    self.gen_add_sync(False)
    self.gen_add_code_line("// column `lane` of dc/dq and of dc/dqd: rows k <= lane use this lane's t3, t2 with S_k; rows k > lane use t1_k, t4_k with this lane's Pdd, Pd, S")
    L = self.tip_L
    self.gen_add_code_line("T dq[%d], dqd[%d]; // rows of this lane's own chain (every other row of the column is structurally zero)" % (L, L))
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int k = 0; k < %d; k++) {" % L, True)
    self.gen_add_code_line("T g[16];")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 16; r++) { g[r] = %s[16*(base + k) + r]; }" % s_G)
    self.gen_add_code_line("const T up_q = g[0]*t3[0] + g[1]*t3[1] + g[2]*t3[2] + g[3]*t3[3] + g[4]*t3[4] + g[5]*t3[5];")
    self.gen_add_code_line("const T up_d = g[0]*t2[0] + g[1]*t2[1] + g[2]*t2[2] + g[3]*t2[3] + g[4]*t2[4] + g[5]*t2[5];")
    self.gen_add_code_line("const T lo_q = g[6]*Pdd[0] + g[7]*Pdd[1] + g[8]*Pdd[2] + g[9]*Pdd[3] + g[10]*Pdd[4] + g[11]*Pdd[5] + g[12]*Pd[0] + g[13]*Pd[1] + g[14]*Pd[2];")
    self.gen_add_code_line("const T lo_d = static_cast<T>(2)*(g[6]*Pd[0] + g[7]*Pd[1] + g[8]*Pd[2] + g[9]*Pd[3] + g[10]*Pd[4] + g[11]*Pd[5]) + g[12]*S[0] + g[13]*S[1] + g[14]*S[2];")
    self.gen_add_code_line("dq[k] = (k <= pos) ? up_q : lo_q;")
    self.gen_add_code_line("dqd[k] = ((k <= pos) ? up_d : lo_d) + ((k == pos) ? Lc[10] : static_cast<T>(0)); // + damping on the diagonal (oracle _test.py:486)")
    self.gen_add_end_control_flow()
    if minv is None:
        self.gen_add_code_line("if (lane < %d) {" % n, True)
        _own_rows(self, dst + "[lane*" + str(n) + " + %d]", "dq[%d]")
        _own_rows(self, dst + "[(" + str(n) + " + lane)*" + str(n) + " + %d]", "dqd[%d]")
        self.gen_add_end_control_flow()
        return
    self.gen_add_code_line("// finally df/du = -Minv*dc/du for the two columns this lane owns (Minv is read wave-uniformly from LDS)")
    self.gen_add_code_line("if (lane < %d) {" % n, True)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int row = 0; row < %d; row++) {" % n, True)
    self.gen_add_code_line("T vq = static_cast<T>(0), vd = static_cast<T>(0);")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int i = 0; i < %d; i++) { const T mi = %s[row*%d + base + i]; vq += mi*dq[i]; vd += mi*dqd[i]; }" % (L, minv, ld))
    self.gen_add_code_line("%s[lane*%d + row] = -vq; %s[(%d + lane)*%d + row] = -vd;" % (dst, n, dst, n, n))
    self.gen_add_end_control_flow()
    self.gen_add_end_control_flow()


def _emit_link_constants_load(self):
    n = self.model.n
    self.gen_add_code_line("T Lc[12]; // this lane's link constants: Ic (6, about the centre of mass), c (3), m, damping, axis")
    self.gen_add_code_line("{ const T *d_L = &grid_model_constants(static_cast<const T *>(nullptr))[%d + 12*lane]; (void)d_robotModel;" % (54 * n))
    self.gen_add_code_line("  #pragma unroll")
    self.gen_add_code_line("  for (int r = 0; r < 12; r++) { Lc[r] = d_L[r]; } }")
    L, nseg = self.tip_L, self.tip_nseg
    if nseg == 1:
        self.gen_add_code_line("const int base = 0, pos = lane; // one chain: position in the chain == lane")
    else:
        self.gen_add_code_line("const int seg = lane / %d; // %d chains of %d joints side by side: chain of this lane's joint, its first joint, the position inside it" % (L, nseg, L))
        self.gen_add_code_line("const int base = %d*((seg < %d) ? seg : %d), pos = lane - base; // (lanes without a joint follow the last chain with pos >= %d: they match no step)" % (L, nseg, nseg - 1, L))


def _emit_ldl_factor(self, A="A", U="Uf", rd="rd", probe=False):
    """In-register U D U^T factorisation of the symmetric n x n matrix A (upper triangle A[i][j], i <= j), eliminating the tip joint
    first: A = Uf diag(1/rd) Uf^T with Uf unit upper triangular.  Wave-uniform (every lane factors the same matrix): n reciprocals,
    n(n-1)/2 multiplies, (n-1)n(n+1)/6 FMAs and no cross-lane traffic.  Diagonal scaling does not affect an unpivoted symmetric
    factorisation, so the badly scaled joint-space inertia (1e-3 at the wrist, 10 at the base) is harmless.
    (Forests: n is the chain length and the matrix is the lane's own diagonal block.)"""
    n = self.tip_L
    for k in range(n - 1, 0, -1):
        R32 = (lambda e: "static_cast<T>(static_cast<float>(%s))" % e) if probe else (lambda e: e)
        self.gen_add_code_line("const T %s%d = %s;" % (rd, k, R32("grid_rcp(%s%d_%d)" % (A, k, k))))
        self.gen_add_code_line(" ".join("const T %s%d_%d = %s;" % (U, i, k, R32("%s%d_%d*%s%d" % (A, i, k, rd, k))) for i in range(k)))
        for j in range(k):
            self.gen_add_code_line(" ".join("%s%d_%d -= %s%d_%d*%s%d_%d;" % (A, i, j, U, i, k, A, j, k) for i in range(j + 1)))
    self.gen_add_code_line("const T %s0 = grid_rcp(%s0_0);" % (rd, A))


def _emit_ldl_solve(self, b, U="Uf", rd="rd"):
    """b <- A^-1 b for the register vector b[n] with the factors of _emit_ldl_factor: Uf y = b, z = y/D, Uf^T x = z."""
    n = self.tip_L
    for k in range(n - 1, 0, -1):
        self.gen_add_code_line(" ".join("%s[%d] -= %s%d_%d*%s[%d];" % (b, i, U, i, k, b, k) for i in range(k)))
    self.gen_add_code_line(" ".join("%s[%d] *= %s%d;" % (b, k, rd, k) for k in range(n)))
    for k in range(1, n):
        self.gen_add_code_line("%s[%d] -= %s;" % (b, k, " + ".join("%s%d_%d*%s[%d]" % (U, i, k, b, i) for i in range(k))))


def gen_forward_dynamics_gradient_inner_tip(self, use_thread_group=False, use_qdd_Minv_input=False, with_so=False):
    """The fused inner of the tip-frame path.
    with_so (u-input form only): forward_dynamics_gradient_inner_tip_so - what fdsva_so runs.  At the end of the gradient every per-joint quantity the
    second-order inverse-dynamics derivatives are built from is in this lane's registers, evaluated at qdd = FD(q, qd, u): S, Pd, Pdd, the composites
    I^C, B^C and t1..t4 (= T1, T4, T3, -T2 of algorithms/_idsva_so.py); the function goes straight on to the idsva_so main loops instead of
    idsva_so_device deriving frames, velocities and composites a second time (the reference composes the two calls, algorithms/_fdsva_so.py:147-156).
    (qdd, Minv)-input form: frame chain, link setup, assembly of dc/du, product with the caller's M^-1.
    u-input form: additionally the joint-space inertia itself comes from the tip-frame composites, M[k][j] = S_k . (I^C_j S_j) for k <= j,
    and is never inverted: every lane factors it (U D U^T, wave-uniform registers) and solves for tau - c and for its own two columns
    of dc/du.  No link-local transforms, no articulated inertias, no M^-1 sweeps."""
    m = self.model
    n = m.n
    assert not (with_so and use_qdd_Minv_input)
    name = "forward_dynamics_gradient_inner_tip" + ("_qdd_minv" if use_qdd_Minv_input else "") + ("_so" if with_so else "")
    notes = ["serial revolute chains only; all link quantities are expressed in the frame of the tip link (see the module notes of",
             "algorithms/_tip_frame_gradient.py); same results as direct_minv_inner + inverse_dynamics_inner + inverse_dynamics_gradient_inner",
             "s_df_du receives -Minv*dc/du in the device layout [col*n + row]; the caller must grid_wave_sync() before other lanes read it"]
    params = ["s_df_du is a pointer to LDS for the final result of size 2*NUM_JOINTS*NUM_JOINTS = " + str(2 * n * n),
              "s_qd is the vector of joint velocities in LDS"]
    if use_qdd_Minv_input:
        params += ["s_qdd is the vector of joint accelerations in LDS", "s_Minv is the dense symmetric inverse mass matrix in LDS (leading dimension GRID_MINV_LD)",
                   "s_X is this solve's compact X(q) storage; it is overwritten by the per-joint hand-off records"]
        sig = "T *s_df_du, const T *s_qd, const T *s_qdd, const T *s_Minv, T *s_X, const robotModel<T> *d_robotModel, const T gravity, const int lane"
    else:
        params += ["s_u is the vector of joint input torques in LDS", "s_X is this solve's compact X(q) storage (only the rotation blocks are read)",
                   "s_G is LDS scratch for the per-joint hand-off records (16 values per joint)", "s_M is LDS scratch for the joint-space inertia matrix (leading dimension GRID_MINV_LD)"]
        params += ["s_qdd_out, s_Minv_out (optional, after lane): where to leave qdd = FD(q, qd, u) and the dense M^-1 (leading dimension GRID_MINV_LD; may be s_M) - what fdsva_so needs besides the gradient"]
        sig = "T *s_df_du, const T *s_qd, const T *s_u, const T *s_X, T *s_G, T *s_M, const robotModel<T> *d_robotModel, const T gravity, const int lane, T *s_qdd_out = nullptr, T *s_Minv_out = nullptr"
        if with_so:
            params += ["so is the idsva_so record of this solve (the compact staging record where GRID_SO_COMPACT is 1, else the dense 4*NUM_JOINTS^3 record; LDS or global memory);",
                       "   the tensors are those of inverse dynamics at qdd = FD(q, qd, u); it may overlap s_X, s_G, s_M, s_qd and s_u (all dead by the time it is written), not s_df_du / s_Minv_out / s_rec",
                       "active is false for lane groups without a solve (they compute but do not store through so)"]
            params += ["s_rec is LDS for the per-joint records [S | Pd | Pdd] of the second-order loops, 20 values per joint (may be s_X)"]
            sig = sig.replace(" = nullptr", "") + ", T *so, T *s_rec, const bool active"  # (no __restrict__: so may overlap the gradient's dead working set)
    params += ["d_robotModel is the pointer to the initialized model specific helpers on the GPU", "gravity is the gravity constant",
               "lane is the caller's lane index inside the solve's lane group"]
    self.gen_add_func_doc("Computes the gradient of forward dynamics in the tip frame", notes, params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void %s(%s) {" % (name, sig), True)
    ts_mode = (not use_qdd_Minv_input) and self.tuning["debug_stop"] == 20  # profiling build: per-wave cycle stamps at the phase boundaries replace the first 8 outputs

    def TS(i):
        if ts_mode:
            self.gen_add_code_line("asm volatile(\"s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_memtime %%0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(grid_ts[%d]) : : \"memory\");" % i)

    if ts_mode:
        self.gen_add_code_line("unsigned long long grid_ts[8];")
    TS(0)
    _emit_link_constants_load(self)
    _emit_chain_decls(self)
    chain_lds = (not use_qdd_Minv_input) and self.tip_nseg == 1 and self.tuning["tip_chain"] == "lds"
    L = self.tip_L
    bf = (not use_qdd_Minv_input) and self.tip_jB is not None
    RS = self.tip_rec if not use_qdd_Minv_input else 16  # values per hand-off record
    if self.tuning["debug_stop"] == 21:  # timing ablation (wrong results): no frame chain at all - what is the walk worth?
        self.gen_add_code_line("gvec[0] = gvec[1] = static_cast<T>(0); gvec[2] = gravity; // (ablation: every frame is the identity)")
    else:
        for i in range(L - 1, -1, -1):
            _chain_step(self, i, "s_G" if chain_lds else None, base_family=bf)
    if chain_lds:
        self.gen_add_sync(use_thread_group)
        self.gen_add_code_line("if (lane < %d) { // this lane's own frame (lanes without a joint keep the identity; their link constants are zero)" % (n - 1), True)
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 9; r++) { myR[r] = s_G[16*lane + r]; }")
        self.gen_add_code_line("myp[0] = s_G[16*lane + 9]; myp[1] = s_G[16*lane + 10]; myp[2] = s_G[16*lane + 11];")
        self.gen_add_end_control_flow()
        self.gen_add_sync(use_thread_group)  # (orders the record writes below behind these reads for lane-per-thread execution models; free on the GPU)
    if use_qdd_Minv_input:
        self.gen_add_sync(use_thread_group)  # every lane is done with s_X before the hand-off records overwrite it
        _emit_link_setup(self)
        self.gen_add_code_line("const T qdd = (lane < %d) ? s_qdd[lane] : static_cast<T>(0);" % n)
        _emit_bias(self, True)
        _emit_assembly(self, s_G="s_X")
        self.gen_add_end_function()
        return
    stop = int(self.tuning["debug_stop"])  # timing ablation only (results are wrong when set): 5 = chain, 6 = + link setup/bias/record, 7 = + M, factorisation, qdd
    ld = self.minv_ld
    if stop == 5:
        self.gen_add_code_line("if (lane < %d) { s_df_du[lane] = myR[0] + myp[0] + gvec[0] + Lc[0]; }" % n)
        self.gen_add_end_function()
        return
    if ts_mode:
        self.gen_add_code_line("grid_pin(myR[0]); grid_pin(myp[0]); grid_pin(gvec[0]);")
    TS(1)
    _probe(self, "chain", "myR:9", "myp:3", "gvec:3")
    _emit_link_setup(self, base_family=bf)
    _probe(self, "link", "S:6", "I:10")
    _probe(self, "vel", "v:6", "Pd:6")
    _emit_bias(self, False)
    _probe(self, "comp_I", "IC:10")
    _probe(self, "comp_BF", "BC:12", "fC:6", "a:6")
    self.gen_add_code_line("// everything that does not depend on qdd: t1, t2, t4, the bias force; one hand-off record per joint: [S | t1 | t4 | tau - c]")
    self.gen_add_code_line("T t1[6], t2[6], t4[3];")
    self.gen_add_code_line("grid_rbi_mul(t1, IC, S);")
    self.gen_add_code_line("grid_bmul(t2, BC, S); grid_rbi_mul_peq(t2, IC, Pd, static_cast<T>(2));")
    self.gen_add_code_line("grid_btmul(t4, BC, S);")
    _probe(self, "t1", "t1:6")
    _probe(self, "t24", "t2:6", "t4:3")
    if bf:
        _emit_base_family(self)
    self.gen_add_code_line("if (lane < %d) {" % n, True)
    self.gen_add_code_line("T *rec = &s_G[%d*lane];" % RS)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 6; r++) { rec[r] = S[r]; rec[6 + r] = t1[r]; }")
    self.gen_add_code_line("rec[12] = t4[0]; rec[13] = t4[1]; rec[14] = t4[2]; rec[15] = s_u[lane] - (grid_dot6(S, fC) + Lc[10]*qd);")
    if bf:
        self.gen_add_code_line("rec[16] = SB[0]; rec[17] = SB[1]; rec[18] = SB[2]; rec[19] = static_cast<T>(0); // linear part of S about pB")
    if "rhs" in tuple(self.tuning.get("round_probe", ())):
        self.gen_add_code_line("rec[15] = static_cast<T>(static_cast<float>(rec[15]));")
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)
    TS(2)
    if stop == 6:
        self.gen_add_code_line("if (lane < %d) { s_df_du[lane] = s_G[%d*lane + 15] + t2[0] + t2[1] + t2[2] + t2[3] + t2[4] + t2[5]; }" % (n, RS))
        self.gen_add_end_function()
        return
    self.gen_add_code_line("// pass 1 over the records: column `lane` of M (rows k <= lane), column `lane` of dc/dqd (complete: it does not depend on qdd),")
    self.gen_add_code_line("// the qdd-independent part of the rows k > lane of dc/dq, and tau - c of every joint")
    Lp = (L + 3) // 4 * 4
    self.gen_add_code_line("T dq[%d], dqd[%d], rhs[%d], Mcol[%d]; // rows of this lane's own chain" % (L, L, L, Lp))
    for k in range(L, Lp):
        self.gen_add_code_line("Mcol[%d] = static_cast<T>(0);" % k)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int k = 0; k < %d; k++) {" % L, True)
    self.gen_add_code_line("T g[%d];" % RS)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < %d; r++) { g[r] = s_G[%d*(base + k) + r]; }" % (RS, RS))
    if bf:
        self.gen_add_code_line("const T l0 = fam ? g[16] : g[3], l1 = fam ? g[17] : g[4], l2 = fam ? g[18] : g[5]; // linear part of S_k about this column's reference point")
        self.gen_add_code_line("const T mkj  = g[0]*t1m[0] + g[1]*t1m[1] + g[2]*t1m[2] + l0*t1m[3] + l1*t1m[4] + l2*t1m[5];")
    else:
        self.gen_add_code_line("const T mkj  = g[0]*t1[0] + g[1]*t1[1] + g[2]*t1[2] + g[3]*t1[3] + g[4]*t1[4] + g[5]*t1[5];")
    self.gen_add_code_line("const T up_d = g[0]*t2[0] + g[1]*t2[1] + g[2]*t2[2] + g[3]*t2[3] + g[4]*t2[4] + g[5]*t2[5];")
    self.gen_add_code_line("const T lo_d = static_cast<T>(2)*(g[6]*Pd[0] + g[7]*Pd[1] + g[8]*Pd[2] + g[9]*Pd[3] + g[10]*Pd[4] + g[11]*Pd[5]) + g[12]*S[0] + g[13]*S[1] + g[14]*S[2];")
    self.gen_add_code_line("dq[k] = g[12]*Pd[0] + g[13]*Pd[1] + g[14]*Pd[2];")
    self.gen_add_code_line("dqd[k] = ((k <= pos) ? up_d : lo_d) + ((k == pos) ? Lc[10] : static_cast<T>(0)); // + damping on the diagonal (oracle _test.py:486)")
    self.gen_add_code_line("rhs[k] = g[15];")
    self.gen_add_code_line("Mcol[k] = mkj; // (rows k > lane are never read)")
    self.gen_add_end_control_flow()
    _probe(self, "M", "Mcol:%d" % L)
    _probe(self, "pass1", "dq:%d" % L, "dqd:%d" % L)
    self.gen_add_code_line("if (lane < %d) {" % n, True)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int k = 0; k < %d; k++) { s_M[%d*lane + k] = Mcol[k]; }" % (Lp, ld))
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)
    TS(3)
    self.gen_add_code_line("// the joint-space inertia (of this lane's chain), uniform over the chain's lanes: upper triangle A<i>_<j> = M[i][j], i <= j  (column j was written by the lane of joint j)")
    for j in range(L):
        self.gen_add_code_line(" ".join("T A%d_%d = s_M[%d*(base + %d) + %d];" % (i, j, ld, j, i) for i in range(j + 1)))
    if "Mread" in tuple(self.tuning.get("round_probe", ())):
        for j in range(L):
            self.gen_add_code_line(" ".join("A%d_%d = static_cast<T>(static_cast<float>(A%d_%d));" % (i, j, i, j) for i in range(j + 1)))
    _emit_ldl_factor(self, probe="factor" in tuple(self.tuning.get("round_probe", ())))
    self.gen_add_code_line("// qdd = M^-1 (tau - c); this lane keeps the entry of its own joint")
    _emit_ldl_solve(self, "rhs")
    sel = "rhs[0]"
    for k in range(1, L):
        sel = "((pos == %d) ? rhs[%d] : %s)" % (k, k, sel)
    if "qdd" in tuple(self.tuning.get("round_probe", ())):
        sel = "static_cast<T>(static_cast<float>(%s))" % sel
    self.gen_add_code_line("const T qdd = (lane < %d) ? %s : static_cast<T>(0);" % (n, sel))
    if ts_mode:
        self.gen_add_code_line("{ T qp = qdd; grid_pin(qp); }")
    TS(4)
    if stop == 7:
        self.gen_add_code_line("if (lane < %d) { s_df_du[lane] = qdd + %s; }" % (n, " + ".join("dq[%d] + dqd[%d]" % (k, k) for k in range(L))))
        self.gen_add_end_function()
        return
    self.gen_add_code_line("// the acceleration-dependent parts: a += sum over the ancestors of S_k qdd_k, f^C += sum over the subtree of I_k da_k")
    self.gen_add_code_line("{", True)
    self.gen_add_code_line("T da[6], Ida[6];")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 6; r++) { da[r] = S[r]*qdd; }")
    self.gen_add_code_line("grid_prefix_sum(da, mku);")
    self.gen_add_code_line("grid_rbi_mul(Ida, I, da); grid_suffix_sum(Ida, mkd);")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 6; r++) { a[r] += da[r]; fC[r] += Ida[r]; }")
    self.gen_add_end_control_flow()
    self.gen_add_code_line("T Pdd[6]; grid_mxm(Pdd, a, S); grid_mxm_peq(Pdd, v, Pd);")
    self.gen_add_code_line("T t3[6]; grid_bmul(t3, BC, Pd); grid_rbi_mul_peq(t3, IC, Pdd, static_cast<T>(1)); grid_fxv_peq(t3, S, fC);")
    _probe(self, "t3", "t3:6", "Pdd:6")
    self.gen_add_code_line("// pass 2 over the records: column `lane` of dc/dq")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int k = 0; k < %d; k++) {" % L, True)
    self.gen_add_code_line("T g[12];")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 12; r++) { g[r] = s_G[%d*(base + k) + r]; }" % RS)
    self.gen_add_code_line("const T up_q = g[0]*t3[0] + g[1]*t3[1] + g[2]*t3[2] + g[3]*t3[3] + g[4]*t3[4] + g[5]*t3[5];")
    self.gen_add_code_line("const T lo_q = g[6]*Pdd[0] + g[7]*Pdd[1] + g[8]*Pdd[2] + g[9]*Pdd[3] + g[10]*Pdd[4] + g[11]*Pdd[5] + dq[k];")
    self.gen_add_code_line("dq[k] = (k <= pos) ? up_q : lo_q;")
    self.gen_add_end_control_flow()
    _probe(self, "pass2", "dq:%d" % L)
    self.gen_add_code_line("// df/du = -M^-1 dc/du for the two columns this lane owns")
    _emit_ldl_solve(self, "dq")
    _emit_ldl_solve(self, "dqd")
    if ts_mode:
        self.gen_add_code_line("grid_pin(dq[0]); grid_pin(dqd[0]);")
    TS(5)
    self.gen_add_code_line("if (lane < %d) {" % n, True)
    if self.tip_nseg == 1:
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int row = 0; row < %d; row++) { s_df_du[lane*%d + row] = -dq[row]; s_df_du[(%d + lane)*%d + row] = -dqd[row]; }" % (n, n, n, n))
    else:
        _own_rows(self, "s_df_du[lane*" + str(n) + " + %d]", "-dq[%d]")
        _own_rows(self, "s_df_du[(" + str(n) + " + lane)*" + str(n) + " + %d]", "-dqd[%d]")
    self.gen_add_end_control_flow()
    if not use_qdd_Minv_input:
        ld = self.minv_ld
        self.gen_add_code_line("if (s_qdd_out != nullptr && lane < %d) { s_qdd_out[lane] = qdd; }" % n)
        self.gen_add_code_line("if (s_Minv_out != nullptr) { // row (= column) `lane` of M^-1 from the register factors: one more solve, of the unit vector of this lane's joint", True)
        self.gen_add_code_line("T e[%d];" % L)
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int k = 0; k < %d; k++) { e[k] = (k == pos) ? static_cast<T>(1) : static_cast<T>(0); }" % L)
        _emit_ldl_solve(self, "e")
        self.gen_add_sync(use_thread_group)  # (s_Minv_out may be the storage M was read from: every lane has read it by now)
        self.gen_add_code_line("if (lane < %d) {" % n, True)
        if self.tip_nseg > 1:
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int k = 0; k < %d; k++) { s_Minv_out[lane*%d + k] = static_cast<T>(0); } // (block diagonal over the chains)" % (n, ld))
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int k = 0; k < %d; k++) { s_Minv_out[lane*%d + base + k] = e[k]; }" % (L, ld))
        self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
    if with_so:
        from ._idsva_so import _so_emit_balanced_main_dots
        compact = self.gen_idsva_so_compact()
        A = self.gen_add_code_line
        A("//")
        A("// second-order derivatives of inverse dynamics at this qdd (algorithms/_idsva_so.py): everything they are built from is in registers")
        A("//")
        A("T T1[6], T2[3], T3[6], T4[6], ICPd[6];")
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { T1[r] = t1[r]; T3[r] = t3[r]; T4[r] = t2[r]; }")
        A("T2[0] = -t4[0]; T2[1] = -t4[1]; T2[2] = -t4[2];")
        A("grid_rbi_mul(ICPd, IC, Pd);")
        self.gen_add_sync(use_thread_group)  # (every lane is done with X(q) and with the records of the gradient)
        A("{ // (from here on s_X names the records of the second-order loops)", True)
        A("T *s_X = s_rec;")
        A("if (lane < %d) { // record of joint `lane`: [S | Pd | Pdd]" % n, True)
        A("T *rec = &s_X[20*lane];")
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { rec[r] = S[r]; rec[6 + r] = Pd[r]; rec[12 + r] = Pdd[r]; }")
        self.gen_add_end_control_flow()
        if compact:
            A("if (lane == 0) { so[%d] = static_cast<T>(0); } // the ZERO slot: what the structurally zero entries of dM_dq expand from" % self.gen_idsva_so_compact_layout()["ZERO"])
        else:
            A("{ // structurally zero entries of dM_dq: dM_ik/dq_j with j <= min(i, k); the owner lane (largest index) writes them first")
            A("  T *mq = so + %d; const int c = lane;" % (3 * n ** 3))
            A("  #pragma unroll 1")
            A("  for (int b = 0; b < %d; b++) { for (int e = 0; e <= b; e++) { if (active && c < %d && b <= c) { mq[(c*%d + e)*%d + b] = static_cast<T>(0); mq[(b*%d + e)*%d + c] = static_cast<T>(0); } } } }" % (n, n, n, n, n, n))
        self.gen_add_sync(use_thread_group)
        _so_emit_balanced_main_dots(self, tree=False, compact=compact)
        self.gen_add_end_control_flow()
    if ts_mode:
        self.gen_add_sync(use_thread_group)
        TS(6)
        self.gen_add_code_line("if (lane == 0) { // cycles since the inner function was entered at each boundary; [7] = low 24 bits of the entry stamp, [8] = hardware id")
        self.gen_add_code_line("    for (int i = 1; i < 7; i++) { s_df_du[i] = static_cast<T>(static_cast<float>(grid_ts[i] - grid_ts[0])); }")
        self.gen_add_code_line("    s_df_du[0] = static_cast<T>(0); s_df_du[7] = static_cast<T>(static_cast<float>(grid_ts[0] & 0xFFFFFFull));")
        self.gen_add_code_line("    s_df_du[8] = static_cast<T>(static_cast<float>(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) & 0xFFFFFF)); }")
    self.gen_add_end_function()


def gen_forward_dynamics_gradient_inner_tip_function_call(self, use_thread_group=False, use_qdd_Minv_input=False, s_df_du_name="s_df_du"):
    if use_qdd_Minv_input:
        self.gen_add_code_line("forward_dynamics_gradient_inner_tip_qdd_minv<T>(%s, s_qd, s_qdd, s_Minv, s_X, d_robotModel, gravity, lane);" % s_df_du_name)
    else:
        fused = "s_qdd_out, s_Minv_out" if not getattr(self, "branch_frame", False) else "nullptr, nullptr"  # (the device function has these parameters where its inner is this one)
        self.gen_add_code_line("forward_dynamics_gradient_inner_tip<T>(%s, s_qd, s_u, s_X, s_U, s_Minv, d_robotModel, gravity, lane, %s); // hand-off records in the U|T scratch, M in the M^-1 slot" % (s_df_du_name, fused))


def _emit_force_only(self, with_qdd):
    """a and the composite force f^C only (no Coriolis matrix): what RNEA itself needs."""
    self.gen_add_code_line("T a[6], fC[6];")
    self.gen_add_code_line("#pragma unroll")
    if with_qdd:
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { a[r] = Pd[r]*qd + S[r]*qdd; }")
    else:
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { a[r] = Pd[r]*qd; }")
    self.gen_add_code_line("grid_prefix_sum(a, mku);")
    self.gen_add_code_line("a[3] += gvec[0]; a[4] += gvec[1]; a[5] += gvec[2];")
    self.gen_add_code_line("{ T Iv[6]; grid_rbi_mul(Iv, I, v); grid_rbi_mul(fC, I, a); grid_fxv_peq(fC, v, Iv); }")
    self.gen_add_code_line("grid_suffix_sum(fC, mkd); // force transmitted across joint j: sum over the links j..n-1")


def _emit_mass_matrix_factor(self, use_thread_group, rhs_expr=None):
    """IC -> t1 = I^C S, record [S | rhs | . | S lin about pB], M column by dots, hand-off, wave-uniform factorisation (leaves Uf*, rd*; rhs[] if asked)."""
    n = self.model.n
    L = self.tip_L
    Lp = (L + 3) // 4 * 4
    ld = self.minv_ld
    bf = self.tip_jB is not None
    self.gen_add_code_line("T t1[6]; grid_rbi_mul(t1, IC, S);")
    if bf:
        _emit_base_family(self)
    self.gen_add_code_line("if (lane < %d) {" % n, True)
    self.gen_add_code_line("T *rec = &s_G[16*lane];")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 6; r++) { rec[r] = S[r]; }")
    if rhs_expr is not None:
        self.gen_add_code_line("rec[6] = %s;" % rhs_expr)
    if bf:
        self.gen_add_code_line("rec[8] = SB[0]; rec[9] = SB[1]; rec[10] = SB[2]; // linear part of S about pB")
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)
    self.gen_add_code_line("T Mcol[%d]%s;" % (Lp, (", rhs[%d]" % L) if rhs_expr is not None else ""))
    for k in range(L, Lp):
        self.gen_add_code_line("Mcol[%d] = static_cast<T>(0);" % k)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int k = 0; k < %d; k++) { // M[k][lane] = S_k . (I^C_lane S_lane), rows k <= lane of the lane's own chain" % L, True)
    nr = 12 if bf else 8
    self.gen_add_code_line("T g[%d];" % nr)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < %d; r++) { g[r] = s_G[16*(base + k) + r]; }" % nr)
    if bf:
        self.gen_add_code_line("const T l0 = fam ? g[8] : g[3], l1 = fam ? g[9] : g[4], l2 = fam ? g[10] : g[5]; // linear part of S_k about this column's reference point")
        self.gen_add_code_line("Mcol[k] = g[0]*t1m[0] + g[1]*t1m[1] + g[2]*t1m[2] + l0*t1m[3] + l1*t1m[4] + l2*t1m[5];")
    else:
        self.gen_add_code_line("Mcol[k] = g[0]*t1[0] + g[1]*t1[1] + g[2]*t1[2] + g[3]*t1[3] + g[4]*t1[4] + g[5]*t1[5];")
    if rhs_expr is not None:
        self.gen_add_code_line("rhs[k] = g[6];")
    self.gen_add_end_control_flow()
    self.gen_add_code_line("if (lane < %d) {" % n, True)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int k = 0; k < %d; k++) { s_M[%d*lane + k] = Mcol[k]; }" % (Lp, ld))
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)
    for j in range(L):
        self.gen_add_code_line(" ".join("T A%d_%d = s_M[%d*(base + %d) + %d];" % (i, j, ld, j, i) for i in range(j + 1)))
    _emit_ldl_factor(self)


def _tip_inner_header(self, name, doc, notes, params, sig, with_gravity=True, base_family=False):
    self.gen_add_func_doc(doc, ["serial revolute chains: computed in the frame of the tip link (algorithms/_tip_frame_gradient.py)"] + notes,
                          params + ["d_robotModel is the pointer to the initialized model specific helpers on the GPU", "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void %s(%s, const robotModel<T> *d_robotModel, const int lane) {" % (name, sig), True)
    _emit_link_constants_load(self)
    _emit_chain_decls(self)
    for i in range(self.tip_L - 1, -1, -1):
        _chain_step(self, i, None, with_gravity, base_family=base_family)


def gen_inverse_dynamics_inner_tip(self, use_thread_group=False):
    """c = S . f^C + damping*qd per joint (RNEA, oracle /root/reference/_test.py:5-115), lane j writes s_c[j]."""
    n = self.model.n
    _tip_inner_header(self, "inverse_dynamics_inner_tip", "Compute the RNEA (Recursive Newton-Euler Algorithm)", ["lane j writes s_c[j]; the caller must grid_wave_sync() before other lanes read it"],
                      ["s_c is the output vector of joint torques in LDS", "s_qd is the vector of joint velocities in LDS",
                       "s_qdd is the vector of joint accelerations in LDS, or nullptr for qdd = 0", "s_X is this solve's compact X(q) storage", "gravity is the gravity constant"],
                      "T *s_c, const T *s_qd, const T *s_qdd, const T *s_X, const T gravity")
    _emit_link_setup(self)
    self.gen_add_code_line("const T qdd = (s_qdd != nullptr && lane < %d) ? s_qdd[lane] : static_cast<T>(0);" % n)
    _emit_force_only(self, True)
    self.gen_add_code_line("if (lane < %d) { s_c[lane] = grid_dot6(S, fC) + Lc[10]*qd; }" % n)
    self.gen_add_end_function()


def gen_inverse_dynamics_gradient_inner_tip(self, use_thread_group=False):
    """dc/du (oracle /root/reference/_test.py:229-494): the assembly of the forward-dynamics-gradient path without the M^-1 product."""
    _tip_inner_header(self, "inverse_dynamics_gradient_inner_tip", "Computes the gradient of inverse dynamics",
                      ["lane j writes columns j and n+j of s_dc_du ([col*n + row]); the caller must grid_wave_sync() before other lanes read them"],
                      ["s_dc_du is the output in LDS, 2*NUM_JOINTS*NUM_JOINTS values", "s_qd is the vector of joint velocities in LDS",
                       "s_qdd is the vector of joint accelerations in LDS", "s_X is this solve's compact X(q) storage; it is overwritten by the per-joint hand-off records",
                       "gravity is the gravity constant"],
                      "T *s_dc_du, const T *s_qd, const T *s_qdd, T *s_X, const T gravity")
    self.gen_add_sync(use_thread_group)  # every lane is done with s_X before the hand-off records overwrite it
    _emit_link_setup(self)
    self.gen_add_code_line("const T qdd = (lane < %d) ? s_qdd[lane] : static_cast<T>(0);" % self.model.n)
    _emit_bias(self, True)
    _emit_assembly(self, s_G="s_X", dst="s_dc_du", minv=None)
    self.gen_add_end_function()


def gen_forward_dynamics_inner_tip(self, use_thread_group=False):
    """qdd = M^-1 (u - c) with M from the tip-frame composites, factored in registers (oracle /root/reference/_test.py:498-501)."""
    n = self.model.n
    _tip_inner_header(self, "forward_dynamics_inner_tip", "Computes forward dynamics", ["lane j writes s_qdd[j]; the caller must grid_wave_sync() before other lanes read it"],
                      ["s_qdd is the output vector of joint accelerations in LDS", "s_qd is the vector of joint velocities in LDS", "s_u is the vector of joint input torques in LDS",
                       "s_X is this solve's compact X(q) storage", "s_G is LDS scratch for the per-joint hand-off records (16 values per joint)",
                       "s_M is LDS scratch for the joint-space inertia matrix (leading dimension GRID_MINV_LD)", "gravity is the gravity constant"],
                      "T *s_qdd, const T *s_qd, const T *s_u, const T *s_X, T *s_G, T *s_M, const T gravity", base_family=True)
    _emit_link_setup(self, base_family=True)
    _emit_force_only(self, False)
    self.gen_add_code_line("T IC[10];")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 10; r++) { IC[r] = I[r]; }")
    self.gen_add_code_line("grid_suffix_sum(IC, mkd);")
    _emit_mass_matrix_factor(self, use_thread_group, rhs_expr="s_u[lane] - (grid_dot6(S, fC) + Lc[10]*qd)")
    _emit_ldl_solve(self, "rhs")
    sel = "rhs[0]"
    for k in range(1, self.tip_L):
        sel = "((pos == %d) ? rhs[%d] : %s)" % (k, k, sel)
    self.gen_add_code_line("if (lane < %d) { s_qdd[lane] = %s; }" % (n, sel))
    self.gen_add_end_function()


def gen_direct_minv_inner_tip(self, use_thread_group=False):
    """M^-1 (dense, symmetric): lane j solves M x = e_j with the register factors (oracle /root/reference/_test.py:117-226 gives the same matrix)."""
    n = self.model.n
    ld = self.minv_ld
    _tip_inner_header(self, "direct_minv_inner_tip", "Compute the inverse of the mass matrix (dense, symmetric) into LDS",
                      ["lane j writes row j (= column j); the caller must grid_wave_sync() before other lanes' entries are read"],
                      ["s_Minv is the n x n output in LDS (leading dimension GRID_MINV_LD); it also holds M itself on the way",
                       "s_X is this solve's compact X(q) storage", "s_G is LDS scratch for the per-joint hand-off records (16 values per joint)"],
                      "T *s_Minv, const T *s_X, T *s_G", with_gravity=False, base_family=True)
    self.gen_add_code_line("T *s_M = s_Minv;")
    _emit_link_setup(self, kinematics=False, base_family=True)
    self.gen_add_code_line("T IC[10];")
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int r = 0; r < 10; r++) { IC[r] = I[r]; }")
    self.gen_add_code_line("grid_suffix_sum(IC, mkd);")
    _emit_mass_matrix_factor(self, use_thread_group)
    L = self.tip_L
    self.gen_add_code_line("T x[%d];" % L)
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int k = 0; k < %d; k++) { x[k] = (k == pos) ? static_cast<T>(1) : static_cast<T>(0); }" % L)
    _emit_ldl_solve(self, "x")
    self.gen_add_sync(use_thread_group)  # every lane has read M before the same slot receives M^-1
    self.gen_add_code_line("if (lane < %d) {" % n, True)
    _own_rows(self, "s_Minv[" + str(ld) + "*lane + %d]", "x[%d]")
    self.gen_add_end_control_flow()
    self.gen_add_code_line("(void)gvec; (void)mku;")
    self.gen_add_end_function()


def gen_tip_frame_components(self, use_thread_group=False):
    self.gen_inverse_dynamics_inner_tip(use_thread_group)
    self.gen_inverse_dynamics_gradient_inner_tip(use_thread_group)
    self.gen_forward_dynamics_inner_tip(use_thread_group)
    self.gen_direct_minv_inner_tip(use_thread_group)


def gen_tip_frame_fused_so(self):
    """True where fdsva_so runs the fused inner forward_dynamics_gradient_inner_tip_so (one chain, the balanced dot-product loops, tuning so_fused)."""
    return bool(self.tip_frame and not getattr(self, "branch_frame", False) and self.tip_nseg == 1 and self.tuning["so_fused"] and self.gen_idsva_so_mode() == "chain"
                and self.tuning["so_mapping"] == "balanced" and self.tuning["so_loops"] == "dots" and int(self.tuning["debug_stop"]) in (0, 30, 31, 32))


def gen_tip_frame_gradient(self, use_thread_group=False):
    self.gen_forward_dynamics_gradient_inner_tip(use_thread_group, False)
    self.gen_forward_dynamics_gradient_inner_tip(use_thread_group, True)
    if self.gen_tip_frame_fused_so():
        self.gen_forward_dynamics_gradient_inner_tip(use_thread_group, False, with_so=True)
