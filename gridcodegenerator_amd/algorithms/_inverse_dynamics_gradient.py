"""Analytical RNEA derivatives (dc/dq, dc/dqd), emitter for the HIP/CDNA4 backend.

Mirrors the role of the reference's algorithms/_inverse_dynamics_gradient.py (gen_inverse_dynamics_gradient_inner
:27-775, device :777, kernel :817, host :890) and follows the reference oracle /root/reference/_test.py:229-494
(test_rnea_grad_inner; Carpentier & Mansard, "Analytical Derivatives of Rigid Body Dynamics Algorithms").

Lane mapping.  Each lane of the solve's lane group owns derivative COLUMNS of every link quantity (dv, da, df) as register
6-vectors: with COLS_PER_LANE == 2 lane j owns d/dq_j and d/dqd_j; with COLS_PER_LANE == 1 the first half of the group
owns the d/dq columns and the second half the d/dqd columns.  A column is structurally zero at link k unless its joint is
an ancestor of k or k itself, and zeros propagate through the linear recursions, so the reference's sparsity-compressed
column bookkeeping (helpers/_topology_helpers.py:515-542, _inverse_dynamics_gradient.py:597-651,731-760) and its
shared-memory atomics (:653-655) are not needed.

Register budget.  The oracle's backward sweep (df_parent += X^T df, _test.py:450-470) needs every link's df until the walk
returns - 12 VGPRs per tree level and lane, which spilled to scratch on gfx950.  Instead the extraction
dc[a][col] = S_a^T sum_{k in subtree(a)} (kX_a)^T df_k[col] is evaluated as sum_k J_{k,a} . df_k[col] during the FORWARD walk,
where J_{k,a} = kX_a S_a is exactly the d/dqd_a column of dv_k that the lane owning that column already holds: it is
published through LDS and every lane dots it with its own transient df_k.  Only the single-vector correction
-X^T mxS(S, f_subtree) of column == joint (_test.py:431-440,469-470) still travels up the tree in a light backward sweep.

The wave-uniform link quantities (v, a, f, I v) of the RNEA pass with qdd are recomputed in
registers inside the same depth-first walk; the link forces are parked in LDS between the two sweeps.

Identities used (checked against the oracle by the tests):
  * mxS(S_i, X_i v_parent) == mxS(S_i, v_i)    because v_i = X_i v_parent + S_i qd_i and S_i x S_i = 0
    (the oracle's MxXv, _test.py:308, therefore equals its Mxv, :310; both vanish for root joints);
  * (fx(v) I) dv == fx(v) (I dv)               (the oracle materialises FxvI = fx(v) I, _test.py:403);
  * d v_k / d qd_a == kX_a S_a                  (velocity Jacobian column).
"""


def gen_inverse_dynamics_gradient_inner_temp_mem_size(self):
    return 0


def gen_inverse_dynamics_gradient_kernel_max_temp_mem_size(self):
    return 0


def gen_gradient_slots(self):
    """Column -> (slot, lane) map of the derivative walk.  Columns are ordered by the joint they belong to
    (c = 2*joint + is_qd: q0, qd0, q1, qd1, ...); slot s of lane l holds column c = s*G + l.  A column is structurally zero at
    link k unless its joint is an ancestor of k or k itself, so ordering the columns by joint lets whole slots be skipped
    at generation time for the links where none of their columns is live (chain: the late joints' slot is dead for the
    first links; trees: a slot that only holds another limb's columns is dead for the whole limb)."""
    m = self.model
    n, G, C = m.n, self.lanes_per_solve, self.cols_per_lane
    slots = []
    for s_ in range(C):
        lo, hi = (s_ * G) // 2, min(n, ((s_ + 1) * G) // 2)  # joints whose two columns live in this slot
        if lo >= hi:
            continue
        joints = set(range(lo, hi))
        live = [bool(joints & set(m.ancestors[k] + [k])) for k in range(n)]       # forward quantities (dv, da, df)
        live_w = [bool(joints & set(m.subtree[k])) for k in range(n)]             # backward correction vector w
        slots.append(dict(name="abcd"[s_], index=s_, lo=lo, hi=hi, live=live, live_w=live_w))
    return slots


def gen_inverse_dynamics_gradient_inner_function_call(self, use_thread_group=False, updated_var_names=None, reuse=False):
    outs = ", ".join("dc_" + sl["name"] for sl in self.gen_gradient_slots())
    if reuse and self.reuse_rnea:
        self.gen_add_code_line("inverse_dynamics_gradient_inner_reuse<T>(" + outs + ", s_qd, s_qdd, s_X, s_U, s_J, gravity, lane);")
    else:
        self.gen_add_code_line("inverse_dynamics_gradient_inner<T>(" + outs + ", s_qd, s_qdd, s_X, s_F, s_J, gravity, lane);")


def gen_gradient_outputs_decl(self):
    n = self.model.n
    return "T " + ", ".join("dc_%s[%d]" % (sl["name"], n) for sl in self.gen_gradient_slots()) + ";"


def gen_inverse_dynamics_gradient_inner(self, use_thread_group=False, reuse=False):
    """reuse=True emits inverse_dynamics_gradient_inner_reuse: same walk, but v, I v and fx(v) I v of every link are read from LDS
    (s_F then points at the table forward_dynamics_inner left behind) instead of being recomputed."""
    m = self.model
    n = m.n
    G = self.lanes_per_solve
    slots = self.gen_gradient_slots()
    ZERO, ONE = "static_cast<T>(0)", "static_cast<T>(1)"
    self.gen_add_func_doc("Computes the gradient of inverse dynamics (the derivative columns this lane owns)",
                          ["columns are ordered c = 2*joint + is_qd; slot s of lane l owns column c = s*GRID_LANES_PER_SOLVE + l and returns",
                           "dc_<slot>[i] = d c_i / d u_c  (u_c = q_joint or qd_joint); columns c >= 2*NUM_JOINTS return zeros",
                           "v, a, f of the RNEA pass with qdd are recomputed wave-uniformly inside the same tree walk",
                           "follows /root/reference/_test.py:229-494 incl. the damping term on diag(dc/dqd)"],
                          ["dc_* are the register outputs (one array per slot)", "s_qd is the vector of joint velocities in LDS",
                           "s_qdd is the vector of joint accelerations in LDS", "s_X is this solve's compact X(q) storage",
                           "s_F is LDS scratch for the wave-uniform link forces (6 floats per joint)",
                           "s_J is LDS scratch for the velocity Jacobian columns of the current link (double buffered)",
                           "gravity is the gravity constant", "lane is the caller's lane index inside the solve's lane group"], None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void inverse_dynamics_gradient_inner" + ("_reuse" if reuse else "") + "(" + ", ".join("T (&dc_%s)[%d]" % (sl["name"], n) for sl in slots) +
                           ", const T *s_qd, const T *s_qdd, const T *s_X, T *s_F, T *s_J, const T gravity, const int lane) {", True)
    self.gen_add_code_line("const bool is_qd = (lane & 1) != 0; // odd columns are d/dqd, even columns d/dq")
    for sl in slots:
        self.gen_add_code_line("const int jc_%s = (%d + lane) >> 1; // joint of the column this lane owns in slot %s" % (sl["name"], sl["index"] * G, sl["name"]))
    self.gen_add_code_line("#pragma unroll")
    self.gen_add_code_line("for (int i = 0; i < %d; i++) { %s }" % (n, " ".join("dc_%s[i] = %s;" % (sl["name"], ZERO) for sl in slots)))
    jstride = (6 * n + 3) // 4 * 4

    def pre(k):
        s, p = m.S_index[k], m.parent[k]
        K, P = str(k), str(p)
        jbuf = (m.depth[k] & 1) * jstride  # s_J is double buffered by depth parity: no hand-off sync is needed after the dots
        act = [sl for sl in slots if sl["live"][k]]
        decl = ["v_%s[6]" % K, "a_%s[6]" % K]
        for sl in act:
            decl += ["dv%s_%s[6]" % (sl["name"], K), "da%s_%s[6]" % (sl["name"], K)]
        for sl in slots:
            if sl["live_w"][k]:
                decl.append("w%s_%s[6]" % (sl["name"], K))
        self.gen_add_code_line("T " + ", ".join(decl) + ";")
        for sl in slots:
            if sl["live_w"][k]:
                self.gen_add_code_line("grid_zero6(w%s_%s);" % (sl["name"], K))
        own = [sl for sl in slots if sl["lo"] <= k < sl["hi"]][0]  # the slot that holds joint k's own two columns
        self.gen_add_code_line("const T selq_%s = (jc_%s == %s && !is_qd) ? %s : %s; const T seld_%s = (jc_%s == %s && is_qd) ? %s : %s;" %
                               (K, own["name"], K, ONE, ZERO, K, own["name"], K, ONE, ZERO))
        self.gen_add_code_line("{", True)
        self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*" + K + "]);")
        self.gen_add_code_line("const T qd = s_qd[" + K + "]; const T qdd = s_qdd[" + K + "];")
        self.gen_add_code_line("// wave-uniform link quantities (RNEA with qdd)")
        self.gen_add_code_line("T Xa[6], Mxv[6], MxXa[6], Iv[6];")
        if p == -1:
            self.gen_add_code_line("grid_zero6(v_%s); v_%s[%d] = qd;" % (K, K, s))
            self.gen_add_code_line("grid_zero6(Xa); Xa[3] = X[2]*gravity; Xa[4] = X[5]*gravity; Xa[5] = X[8]*gravity;")
        else:
            self.gen_add_code_line("grid_xmul(v_%s, X, v_%s); v_%s[%d] += qd;" % (K, P, K, s))
            self.gen_add_code_line("grid_xmul(Xa, X, a_%s);" % P)
        self.gen_add_code_line("grid_zero6(Mxv); grid_mxS_peq<T,%d>(Mxv, v_%s, %s);" % (s, K, ONE))
        self.gen_add_code_line("grid_zero6(MxXa); grid_mxS_peq<T,%d>(MxXa, Xa, %s);" % (s, ONE))
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { a_%s[r] = Xa[r] + Mxv[r]*qd; }" % K)
        self.gen_add_code_line("a_%s[%d] += qdd;" % (K, s))
        self.gen_add_code_line("// the link force is wave-uniform and only needed again on the way back up: park it in LDS")
        self.gen_add_code_line("{ T f[6]; grid_imul_%s(Iv, v_%s); grid_imul_%s(f, a_%s); grid_fxv_peq(f, v_%s, Iv);" % (K, K, K, K, K))
        self.gen_add_code_line("  if (lane == 0) {")
        self.gen_add_code_line("      #pragma unroll")
        self.gen_add_code_line("      for (int r = 0; r < 6; r++) { s_F[%d + r] = f[r]; }" % (6 * k))
        self.gen_add_code_line("  } }")
        self.gen_add_code_line("// derivative columns: dv, da (forward recursions; the self terms belong to the columns of joint %s, slot %s)" % (K, own["name"]))
        for sl in act:
            c = sl["name"]
            is_own = sl is own
            parent_live = p != -1 and sl["live"][p]
            if parent_live:
                self.gen_add_code_line("grid_xmul(dv%s_%s, X, dv%s_%s); grid_xmul(da%s_%s, X, da%s_%s);" % (c, K, c, P, c, K, c, P))
                if is_own:
                    self.gen_add_code_line("#pragma unroll")
                    self.gen_add_code_line("for (int r = 0; r < 6; r++) { dv%s_%s[r] += selq_%s*Mxv[r]; da%s_%s[r] += selq_%s*MxXa[r] + seld_%s*Mxv[r]; }" % (c, K, K, c, K, K, K))
                    self.gen_add_code_line("dv%s_%s[%d] += seld_%s;" % (c, K, s, K))
            else:  # no live parent column in this slot: only the self terms (is_own is implied by liveness)
                self.gen_add_code_line("#pragma unroll")
                self.gen_add_code_line("for (int r = 0; r < 6; r++) { dv%s_%s[r] = selq_%s*Mxv[r]; da%s_%s[r] = selq_%s*MxXa[r] + seld_%s*Mxv[r]; }" % (c, K, K, c, K, K, K))
                self.gen_add_code_line("dv%s_%s[%d] += seld_%s;" % (c, K, s, K))
            self.gen_add_code_line("grid_mxS_peq<T,%d>(da%s_%s, dv%s_%s, qd);" % (s, c, K, c, K))
        self.gen_add_code_line("// publish the velocity Jacobian columns J_{k,a} = d v_k / d qd_a (the d/dqd columns of dv) for the proper ancestors")
        if m.ancestors[k]:
            for sl in act:
                if not any(sl["lo"] <= a < sl["hi"] for a in m.ancestors[k]):
                    continue
                c = sl["name"]
                self.gen_add_code_line("if (is_qd && jc_%s < %d) {" % (c, n), True)
                self.gen_add_code_line("#pragma unroll")
                self.gen_add_code_line("for (int r = 0; r < 6; r++) { s_J[%d + 6*jc_%s + r] = dv%s_%s[r]; }" % (jbuf, c, c, K))
                self.gen_add_end_control_flow()
        self.gen_add_code_line("// local df = I da + fx(dv) (I v) + fx(v) (I dv), consumed at once: dc[a] += J_{k,a} . df for a in ancestors(k) + {k}")
        for sl in act:
            c = sl["name"]
            self.gen_add_code_line("T df%s[6]; { T Idv[6]; grid_imul_%s(df%s, da%s_%s); grid_fxv_peq(df%s, dv%s_%s, Iv); grid_imul_%s(Idv, dv%s_%s); grid_fxv_peq(df%s, v_%s, Idv); }" % (c, K, c, c, K, c, c, K, K, c, K, c, K))
        self.gen_add_code_line("// J_{k,k} = S_k: no LDS round trip for the link's own row")
        self.gen_add_code_line(" ".join("dc_%s[%d] += df%s[%d]; grid_pin(dc_%s[%d]);" % (sl["name"], k, sl["name"], s, sl["name"], k) for sl in act))
        if m.ancestors[k]:
            self.gen_add_sync(use_thread_group)
            for a in m.ancestors[k]:
                self.gen_add_code_line("{ T J[6];")
                self.gen_add_code_line("  #pragma unroll")
                self.gen_add_code_line("  for (int r = 0; r < 6; r++) { J[r] = s_J[%d + r]; }" % (jbuf + 6 * a))
                self.gen_add_code_line("  " + " ".join("dc_%s[%d] += grid_dot6(J, df%s); grid_pin(dc_%s[%d]);" % (sl["name"], a, sl["name"], sl["name"], a) for sl in act) + " }")
        self.gen_add_end_control_flow()

    def post(k):
        s, p = m.S_index[k], m.parent[k]
        K, P = str(k), str(p)
        damp = m.damping[k]
        own = [sl for sl in slots if sl["lo"] <= k < sl["hi"]][0]
        self.gen_add_code_line("// corrections that travelled up from the subtree: dc[k][col] -= S_k^T w")
        for sl in slots:
            if sl["live_w"][k]:
                self.gen_add_code_line("dc_%s[%s] -= w%s_%s[%d];" % (sl["name"], K, sl["name"], K, s))
        if damp != 0.0:
            self.gen_add_code_line("dc_%s[%s] += seld_%s*static_cast<T>(%s);" % (own["name"], K, K, repr(float(damp))))
        if p == -1 and k != m.roots[-1]:
            self.gen_add_sync(use_thread_group)  # the next root's subtree reuses the depth-0 J buffer
        if p != -1:
            self.gen_add_sync(use_thread_group)
            self.gen_add_code_line("{", True)
            self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*" + K + "]);")
            self.gen_add_code_line("T f[6], fp[6];")
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { f[r] = s_F[%d + r]; fp[r] = s_F[%d + r]; }" % (6 * k, 6 * p))
            self.gen_add_code_line("// column == joint injects mxS(S, f_subtree); everything is carried to the parent frame")
            self.gen_add_code_line("grid_mxS_peq<T,%d>(w%s_%s, f, selq_%s);" % (s, own["name"], K, K))
            self.gen_add_code_line("grid_xtmul_peq(fp, X, f);")
            for sl in slots:
                if sl["live_w"][k]:
                    self.gen_add_code_line("grid_xtmul_peq(w%s_%s, X, w%s_%s); grid_pin6(w%s_%s);" % (sl["name"], P, sl["name"], K, sl["name"], P))
            self.gen_add_code_line("if (lane == 0) {", True)
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { s_F[%d + r] = fp[r]; }" % (6 * p))
            self.gen_add_end_control_flow()
            self.gen_add_end_control_flow()

    # ---- register-resident variant: for shallow trees the whole derivative walk stays in VGPRs -------------------------------
    # (every link keeps its f and its live df slots until the walk returns: 6 + 6*slots VGPRs per tree level).  The backward
    # sweep df_parent += X^T df (oracle _test.py:450-470) then replaces the Jacobian dots, the J/F hand-offs through LDS and all
    # wave-level syncs of the walk.  Deep trees keep the forward-accumulation form above (O(1) registers per level).
    def pre_reg(k):
        s, p = m.S_index[k], m.parent[k]
        K, P = str(k), str(p)
        act = [sl for sl in slots if sl["live"][k]]
        back = [sl for sl in slots if sl["live"][k] or sl["live_w"][k]]  # slots whose df exists at this link (own or from the subtree)
        decl = ["v_%s[6]" % K, "a_%s[6]" % K, "f_%s[6]" % K]
        for sl in act:
            decl += ["dv%s_%s[6]" % (sl["name"], K), "da%s_%s[6]" % (sl["name"], K)]
        for sl in back:
            decl.append("df%s_%s[6]" % (sl["name"], K))
        self.gen_add_code_line("T " + ", ".join(decl) + ";")
        own = [sl for sl in slots if sl["lo"] <= k < sl["hi"]][0]
        self.gen_add_code_line("const T selq_%s = (jc_%s == %s && !is_qd) ? %s : %s; const T seld_%s = (jc_%s == %s && is_qd) ? %s : %s;" %
                               (K, own["name"], K, ONE, ZERO, K, own["name"], K, ONE, ZERO))
        self.gen_add_code_line("{", True)
        self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*" + K + "]);")
        self.gen_add_code_line("const T qd = s_qd[" + K + "]; const T qdd = s_qdd[" + K + "];")
        self.gen_add_code_line("T Xa[6], Mxv[6], MxXa[6], Iv[6];")
        if reuse:
            self.gen_add_code_line("T b[6];")
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { v_%s[r] = s_F[%d + r]; Iv[r] = s_F[%d + r]; b[r] = s_F[%d + r]; }" % (K, 18 * k, 18 * k + 6, 18 * k + 12))
        if p == -1:
            if not reuse:
                self.gen_add_code_line("grid_zero6(v_%s); v_%s[%d] = qd;" % (K, K, s))
            self.gen_add_code_line("grid_zero6(Xa); Xa[3] = X[2]*gravity; Xa[4] = X[5]*gravity; Xa[5] = X[8]*gravity;")
        else:
            if not reuse:
                self.gen_add_code_line("grid_xmul(v_%s, X, v_%s); v_%s[%d] += qd;" % (K, P, K, s))
            self.gen_add_code_line("grid_xmul(Xa, X, a_%s);" % P)
        self.gen_add_code_line("grid_zero6(Mxv); grid_mxS_peq<T,%d>(Mxv, v_%s, %s);" % (s, K, ONE))
        self.gen_add_code_line("grid_zero6(MxXa); grid_mxS_peq<T,%d>(MxXa, Xa, %s);" % (s, ONE))
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int r = 0; r < 6; r++) { a_%s[r] = Xa[r] + Mxv[r]*qd; }" % K)
        self.gen_add_code_line("a_%s[%d] += qdd;" % (K, s))
        if reuse:
            self.gen_add_code_line("grid_imul_%s(f_%s, a_%s);" % (K, K, K))
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int r = 0; r < 6; r++) { f_%s[r] += b[r]; }" % K)
            self.gen_add_code_line("grid_pin6(f_%s);" % K)
        else:
            self.gen_add_code_line("grid_imul_%s(Iv, v_%s); grid_imul_%s(f_%s, a_%s); grid_fxv_peq(f_%s, v_%s, Iv); grid_pin6(f_%s);" % (K, K, K, K, K, K, K, K))
        for sl in act:
            c = sl["name"]
            is_own = sl is own
            if p != -1 and sl["live"][p]:
                self.gen_add_code_line("grid_xmul(dv%s_%s, X, dv%s_%s); grid_xmul(da%s_%s, X, da%s_%s);" % (c, K, c, P, c, K, c, P))
                if is_own:
                    self.gen_add_code_line("#pragma unroll")
                    self.gen_add_code_line("for (int r = 0; r < 6; r++) { dv%s_%s[r] += selq_%s*Mxv[r]; da%s_%s[r] += selq_%s*MxXa[r] + seld_%s*Mxv[r]; }" % (c, K, K, c, K, K, K))
                    self.gen_add_code_line("dv%s_%s[%d] += seld_%s;" % (c, K, s, K))
            else:
                self.gen_add_code_line("#pragma unroll")
                self.gen_add_code_line("for (int r = 0; r < 6; r++) { dv%s_%s[r] = selq_%s*Mxv[r]; da%s_%s[r] = selq_%s*MxXa[r] + seld_%s*Mxv[r]; }" % (c, K, K, c, K, K, K))
                self.gen_add_code_line("dv%s_%s[%d] += seld_%s;" % (c, K, s, K))
            self.gen_add_code_line("grid_mxS_peq<T,%d>(da%s_%s, dv%s_%s, qd);" % (s, c, K, c, K))
            self.gen_add_code_line("{ T Idv[6]; grid_imul_%s(df%s_%s, da%s_%s); grid_fxv_peq(df%s_%s, dv%s_%s, Iv); grid_imul_%s(Idv, dv%s_%s); grid_fxv_peq(df%s_%s, v_%s, Idv); grid_pin6(df%s_%s); }" %
                                   (K, c, K, c, K, c, K, c, K, K, c, K, c, K, K, c, K))
        for sl in back:
            if sl not in act:
                self.gen_add_code_line("grid_zero6(df%s_%s);" % (sl["name"], K))
        self.gen_add_end_control_flow()

    def post_reg(k):
        s, p = m.S_index[k], m.parent[k]
        K, P = str(k), str(p)
        damp = m.damping[k]
        own = [sl for sl in slots if sl["lo"] <= k < sl["hi"]][0]
        back = [sl for sl in slots if sl["live"][k] or sl["live_w"][k]]
        for sl in back:
            self.gen_add_code_line("dc_%s[%s] = df%s_%s[%d]; grid_pin(dc_%s[%s]);" % (sl["name"], K, sl["name"], K, s, sl["name"], K))
        if damp != 0.0:
            self.gen_add_code_line("dc_%s[%s] += seld_%s*static_cast<T>(%s);" % (own["name"], K, K, repr(float(damp))))
        if p != -1:
            self.gen_add_code_line("{", True)
            self.gen_add_code_line("T X[18]; grid_load_X(X, &s_X[GRID_X_STRIDE*" + K + "]);")
            self.gen_add_code_line("// column == joint picks up -X^T mxS(S, f_subtree); everything is carried to the parent frame")
            self.gen_add_code_line("grid_mxS_peq<T,%d>(df%s_%s, f_%s, -selq_%s);" % (s, own["name"], K, K, K))
            self.gen_add_code_line("grid_xtmul_peq(f_%s, X, f_%s); grid_pin6(f_%s);" % (P, K, P))
            for sl in back:
                self.gen_add_code_line("grid_xtmul_peq(df%s_%s, X, df%s_%s); grid_pin6(df%s_%s);" % (sl["name"], P, sl["name"], K, sl["name"], P))
            self.gen_add_end_control_flow()

    use_regs = self.register_walk
    if use_regs:
        self.gen_add_code_line("(void)s_F; (void)s_J; // the register-resident walk needs no LDS hand-offs" + (" (s_F is the read-only v / I v / fx(v) I v table here)" if reuse else ""))
        self.gen_tree_traversal(pre_reg, post_reg)
    else:
        self.gen_tree_traversal(pre, post)
    self.gen_add_end_function()


def gen_dc_du_to_lds(self, dst="s_dc_du", minv_name=None):
    """Emits the register -> LDS staging of this lane's column(s) in the device layout [col*n + row], col = joint + n*is_qd.
    With minv_name set, the staged values are -Minv * dc (the forward-dynamics gradient) instead of dc itself."""
    n = self.model.n
    G = self.lanes_per_solve
    for sl in self.gen_gradient_slots():
        c = sl["name"]
        self.gen_add_code_line("if (%d + lane < %d) {" % (sl["index"] * G, 2 * n), True)
        self.gen_add_code_line("const int col = ((%d + lane) >> 1) + (((lane & 1) != 0) ? %d : 0);" % (sl["index"] * G, n))
        self.gen_add_code_line("#pragma unroll")
        if minv_name is None:
            self.gen_add_code_line("for (int row = 0; row < %d; row++) { %s[col*%d + row] = dc_%s[row]; }" % (n, dst, n, c))
        else:
            self.gen_add_code_line("for (int row = 0; row < %d; row++) {" % n, True)
            self.gen_add_code_line("T val = static_cast<T>(0);")
            self.gen_add_code_line("#pragma unroll")
            self.gen_add_code_line("for (int i = 0; i < %d; i++) { val += %s[row*%d + i]*dc_%s[i]; }" % (n, minv_name, self.minv_ld, c))
            self.gen_add_code_line("%s[col*%d + row] = -val;" % (dst, n))
            self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()


def gen_inverse_dynamics_gradient_device(self, use_thread_group=False, use_qdd_input=False):
    n = self.model.n
    params = ["s_dc_du is the output in LDS, 2*NUM_JOINTS*NUM_JOINTS values laid out [col*n + row], col in [0,2n) = [d/dq | d/dqd]",
              "s_q is the vector of joint positions in LDS", "s_qd is the vector of joint velocities in LDS"]
    if use_qdd_input:
        params.append("s_qdd is the vector of joint accelerations in LDS")
    params += ["s_work is this solve's LDS workspace of GRID_LDS_PER_SOLVE elements", "d_robotModel is the pointer to the initialized model specific helpers on the GPU",
               "gravity is the gravity constant", "lane is the caller's lane index inside the solve's lane group"]
    self.gen_add_func_doc("Computes the gradient of inverse dynamics: X(q) update + inverse_dynamics_gradient_inner (lane-group cooperative)",
                          ["all lanes of the solve's lane group must call it; s_dc_du is visible to the group on return"], params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void inverse_dynamics_gradient_device(T *s_dc_du, const T *s_q, const T *s_qd, " + ("const T *s_qdd, " if use_qdd_input else "") +
                           "T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane, const int off_sp = GRID_OFF_SP, const int off_qdd = GRID_OFF_QDD) {", True)
    self.gen_add_code_line("T *s_X = &s_work[GRID_OFF_X]; T *s_F = &s_work[GRID_OFF_F]; T *s_J = &s_work[GRID_OFF_J]; (void)off_sp; (void)off_qdd; // (off_sp / off_qdd: path-axis scratch and zero-qdd vector inside s_work; the stand-alone kernel carves a compact slice)")
    if not use_qdd_input:
        self.gen_add_code_line("T *s_qdd = &s_work[off_qdd];")
        self.gen_add_code_line("if (lane < %d) { s_qdd[lane] = static_cast<T>(0); }" % n)
    self.gen_load_update_XImats_helpers_function_call(use_thread_group)
    if self.tip_frame:  # serial revolute chains: assembled in the tip link's frame
        self.gen_add_code_line("(void)s_F; (void)s_J;")
        self.gen_add_code_line("inverse_dynamics_gradient_inner_tip<T>(s_dc_du, s_qd, s_qdd, s_X, gravity, d_robotModel, lane);")
        self.gen_add_sync(use_thread_group)
        self.gen_add_end_function()
        return
    if getattr(self, "branch_components", False):  # branched revolute robots: assembled with every branch in its tip link's frame
        self.gen_add_code_line("(void)s_F; (void)s_J;")
        self.gen_add_code_line("inverse_dynamics_gradient_inner_branch<T>(s_dc_du, s_qd, s_qdd, s_X, &s_work[off_sp], d_robotModel, gravity, lane);")
        self.gen_add_sync(use_thread_group)
        self.gen_add_end_function()
        return
    self.gen_add_code_line(self.gen_gradient_outputs_decl())
    self.gen_inverse_dynamics_gradient_inner_function_call(use_thread_group)
    self.gen_dc_du_to_lds("s_dc_du")
    self.gen_add_sync(use_thread_group)
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
    self.gen_add_code_line("__global__ GRID_LAUNCH_BOUNDS")
    self.gen_add_code_line(func_def, True)
    self.gen_kernel_prologue("ID_DU_LDS_PER_SOLVE")
    self.gen_add_code_lines(["T *s_q_qd = &s_mem[GRID_OFF_IN]; T *s_q = s_q_qd; T *s_qd = &s_q_qd[%d]; T *s_qdd = &s_q_qd[%d];" % (n, 2 * n),
                             "T *s_dc_du = &s_out_all[grp*%d];" % (2 * n * n)])
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
    self.gen_add_code_line("inverse_dynamics_gradient_device<T>(s_dc_du, s_q, s_qd, " + ("s_qdd, " if use_qdd_input else "") + "s_mem, d_robotModel, gravity, lane, ID_DU_OFF_SP, GRID_OFF_IN + %d);" % (2 * n))
    if single_call_timing:
        self.gen_add_end_control_flow()
    if single_call_timing:
        self.gen_kernel_save_result_single_timing("dc_du", 2 * n * n, use_thread_group)
    else:
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
    self.gen_add_code_lines(["if (USE_QDD_FLAG) {hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, ID_DU_LDS_PER_SOLVE, ID_DU_OUT_PER_SOLVE),0,hd_data->d_dc_du,d_in,stride_q_qd,hd_data->d_qdd,d_robotModel,gravity,num_timesteps);}",
                             "else {hipLaunchKernelGGL((" + kern + "),block_dimms,thread_dimms,grid_lds_bytes<T>(thread_dimms, ID_DU_LDS_PER_SOLVE, ID_DU_OUT_PER_SOLVE),0,hd_data->d_dc_du,d_in,stride_q_qd,d_robotModel,gravity,num_timesteps);}",
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
    if self.reuse_rnea:
        self.gen_inverse_dynamics_gradient_inner(use_thread_group, reuse=True)
    self.gen_inverse_dynamics_gradient_device(use_thread_group, use_qdd_input=False)
    self.gen_inverse_dynamics_gradient_device(use_thread_group, use_qdd_input=True)
    for use_qdd in (True, False):
        for timing in (True, False):
            self.gen_inverse_dynamics_gradient_kernel(use_thread_group, use_qdd, timing)
    for mode in (0, 1, 2):
        self.gen_inverse_dynamics_gradient_host(mode)
