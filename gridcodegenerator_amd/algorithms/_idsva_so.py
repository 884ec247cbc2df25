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
Scope: fixed-base robots with revolute joints (gen_idsva_so_mode: chain form, tree form; gen_idsva_so_direct for records beyond the LDS of a CU).
"""
from ._tip_frame_gradient import _chain_step, _emit_bias, _emit_body_terms, _emit_chain_decls, _emit_link_constants_load, _emit_link_inertia, _emit_link_setup


def gen_idsva_so_mode(self):
    """"chain": a single serial chain of revolute joints - the tip-frame formulation (lane scans, no LDS hand-offs in the set-up);
    "tree": any other fixed-base robot with revolute joints - one common (base) frame, the kinematics
    and the subtree composites travel level by level through LDS (reference: get_parent_id tables, algorithms/_idsva_so.py:171-193,264-284,320-340);
    None: not emitted (one derivative column per lane; prismatic joints under the round-2 tuning variants).  Records beyond the LDS of a CU (30 joints: 432 KB per solve - the reference's own kernel keeps
    4 n^3 + 390 n + 36 |pairs| values in shared memory, ~500 KB there, and cannot be launched on any GPU) are written straight to global
    memory: gen_idsva_so_direct()."""
    if getattr(self, "tip_frame", False) and self.tip_nseg == 1:
        return "chain"
    m = self.model
    if self.cols_per_lane != 2:
        return None
    if any(s_ >= 3 for s_ in m.S_index) and not (self.tuning["so_origin"] == "joint" and self.tuning["so_mapping"] == "balanced" and self.tuning["so_loops"] == "dots"):
        return None  # (prismatic joints: only the form that keeps every joint's quantities about its own origin carries the joint type through the records)
    return "tree"


def gen_idsva_so_direct(self):
    """True where the 4 n^3 record of one solve does not fit the LDS staging next to the slice: the kernels then write every entry straight to
    the solve's record in global memory (store-issue bound, see the module docstring - but the only way for large robots), and fdsva_so reads the
    idsva_so tensors back from a global workspace (gridData::d_idsva_so) instead of LDS."""
    n = self.model.n
    if self.gen_idsva_so_mode() is None:
        return False
    return self.tuning["so_direct"] is True or (28 * n + 4 * n + 16 + 4 * n * n * n) * 4 > 150 * 1024


def gen_idsva_so_available(self):
    return self.gen_idsva_so_mode() is not None


def gen_idsva_so_inner_temp_mem_size(self):
    return 0  # one record per joint inside the scratch area (it re-uses the X(q) block)


def gen_idsva_so_rec(self):
    """Values per joint of the scratch area behind q | qd | qdd: 20 = [S | Pd | Pdd] records (chain); the tree form also parks the level-by-level
    hand-off records there: [R(9) p(3) v(6) a(6)] = 24 on the way down, [I^C(10) B^C(12) f^C(6)] = 28 on the way up."""
    return 28 if self.gen_idsva_so_mode() == "tree" else 20


def gen_idsva_so_lds_layout(self):
    """LDS of the idsva_so kernels: a compact per-solve slice [q | qd | qdd (padded) | scratch = X(q) / per-joint records | qdd zeros] and,
    behind the block's slices, the 4 n^3 output tensors of every solve (staged so that the record leaves with coalesced 16-byte stores:
    scattered 4-byte global stores were measured at 4.7x the cost of the arithmetic).  Returns (slice, scratch, staging, threads)."""
    n, G = self.model.n, self.lanes_per_solve
    pad4 = lambda x: (x + 3) // 4 * 4
    scratch = self.gen_idsva_so_rec() * n + pad4(n)
    sl = pad4(3 * n) + scratch
    if (sl // 4) % 2 == 0:
        sl += 4
    stage = 0 if self.gen_idsva_so_direct() else (self.gen_idsva_so_compact_layout()["SIZE"] if self.gen_idsva_so_packed() else 4 * n * n * n)
    # block size: full waves unless fewer lane groups per block let at least a quarter more solves be resident on a CU (quadruped, 32-lane groups, 16 384
    # solves: one solve per block 231 us, two 274; 6-DoF arm, 16-lane groups, 65 536 solves: four per block 135 us, three - 8 % more resident - 180)
    per = (sl + stage) * 4
    resident = lambda g_: min(155 * 1024 // (g_ * per), 8) * g_  # (LDS capacity, and at most 8 waves per CU: the kernels hold more than 168 VGPRs)
    best_g = max(1, 64 // G)
    for g_ in range(max(1, 32 // G), best_g):
        if 4 * resident(g_) >= 5 * resident(best_g):
            best_g = g_
            break
    threads = best_g * G
    while threads > G and (threads // G) * per > 150 * 1024:
        threads -= G
    return sl, scratch, stage, threads


def gen_idsva_so_tree_tables(self):
    """Per-lane constants of the tree form.  floats (16 per lane, appended to grid_model_constants): Ic (6, about the centre of mass), c (3), m,
    damping, axis, then the origin of the joint's frame in its parent's coordinates (3) and a pad.  ints (K = 4 + max children per lane):
    parent, tree level, subtree size, number of children, child ids."""
    m = self.model
    n, G = m.n, self.lanes_per_solve
    Lc = self.gen_tip_frame_link_constants()[:12 * G]
    fl = []
    for j in range(G):
        fl += Lc[12 * j:12 * j + 12]
        fl += [float(x) for x in self.gen_tip_frame_joint_offset(j)] + [0.0] if j < n else [0.0] * 4
    maxc = max(1, max(len(c) for c in m.children))
    K = 4 + maxc
    it = []
    for j in range(G):
        if j < n:
            it += [m.parent[j], m.depth[j], len(m.subtree[j]), len(m.children[j])] + list(m.children[j]) + [-1] * (maxc - len(m.children[j]))
        else:
            it += [-1, -1, 0, 0] + [-1] * maxc
    return fl, it, K, maxc


def gen_idsva_so_inner_function_call(self, use_thread_group=False, use_qdd_input=False, updated_var_names=None, compact=False):
    zb = ", blocks_only" if (self.gen_idsva_so_mode() == "tree" and len(self.model.roots) > 1) else ""
    self.gen_add_code_line("idsva_so_inner" + ("_compact" if compact else "") + "<T>(so, s_qd, s_qdd, s_X, d_robotModel, gravity, lane, active" + zb + ");")


def gen_idsva_so_inner(self, use_thread_group=False, use_qdd_input=False, compact=False):
    if self.gen_idsva_so_mode() == "tree":
        return gen_idsva_so_inner_tree(self, use_thread_group, blocks=compact)
    return gen_idsva_so_inner_chain(self, use_thread_group, compact)


def gen_idsva_so_items(self):
    """Balanced work distribution of the main loops (tuning so_mapping = balanced).  The reference's triples (joint j, ancestor-or-self an, subtree
    member c) are grouped into ITEMS (c, m), m an ancestor-or-self of c: item (c, m) applies the operators of joint c to the vectors of joint m once
    and then covers the triples (j = l, an = m, c) for every l on the path m .. c and (j = m, an = l, c) for every ancestor-or-self l of m.  With
    lane <-> c (the subtree mapping) lane c executes all n^2 (m, l) combinations although only those with m, l <= c concern it: 43 % useful on a
    7-joint chain.  Here every lane gets the same number of items (sorted by their trip counts so that the lanes of a slot run loops of similar
    length) and fetches joint c's composites from lane c with cross-lane shuffles.  Returns (items[lane][slot] = (c, m) or None, slots,
    max path trips per slot, max ancestor trips per slot)."""
    m_ = self.model
    n, G = m_.n, self.lanes_per_solve
    depth = m_.depth
    dots = self.tuning["so_loops"] == "dots"  # (that form handles the step l = m of the ancestor loop before the loop: one trip fewer)
    items = []
    for c in range(n):
        for a in sorted(m_.ancestors[c]) + [c]:
            items.append((c, a, depth[c] - depth[a] + 1, depth[a] + (0 if dots else 1)))
    slots = (len(items) + G - 1) // G
    best = None
    cI, cA, cB = (800, 75, 95) if dots else (450, 130, 200)  # instructions per item / path step / ancestor step (static counts of the emitted bodies)
    # all lanes of a slot run the path loop and the ancestor loop to the slot's longest trip counts: try a few orderings, keep the cheapest packing
    # (the `it[2] - it[3]` orders put the items with long paths and few ancestors together and those with many ancestors together:
    # 7-joint chain, 16 lanes: 10 + 9 steps instead of 12 + 10)
    for key in (lambda it: (-(it[2] + it[3]), -it[2]), lambda it: (-it[2], -it[3]), lambda it: (-it[3], -it[2]), lambda it: (-(it[2] + it[3]), -it[3]),
                lambda it: (-(it[2] - it[3]), -it[2]), lambda it: (-(it[2] - it[3]), -(it[2] + it[3])), lambda it: (it[2] - it[3], -it[3])):
        order = sorted(items, key=lambda it: key(it) + (it[0], it[1]))
        table = [[None] * slots for _ in range(G)]
        tA, tB = [0] * slots, [0] * slots
        for i, it in enumerate(order):
            sl, ln = divmod(i, G)
            table[ln][sl] = (it[0], it[1])
            tA[sl] = max(tA[sl], it[2])
            tB[sl] = max(tB[sl], it[3])
        cost = sum(cI + cA * a + cB * b for a, b in zip(tA, tB))
        if best is None or cost < best[0]:
            best = (cost, table, tA, tB)
    _, table, tA, tB = best
    return table, slots, tA, tB


def gen_idsva_so_compact(self):
    """True where the kernels stage the record of a solve in COMPACT form (tuning so_stage): every value once - d2tau_dq2 and d2tau_dqd2 are
    symmetric in their last two indices, dM_dq in its first and last, and dM_ik/dq_j is structurally zero for j <= min(i, k) - and the dense
    4 n^3 record is gathered from it through the table grid_so_expand when it leaves for global memory.  Serial chains, LDS-staged form:
    7-DoF arm 1 372 -> 936 values per solve (8 resident waves per CU instead of 6) and a third fewer LDS stores in the main loops."""
    want = self.tuning["so_stage"]
    if want not in ("auto", "compact", "dense"):
        raise ValueError("tuning['so_stage'] must be auto, compact or dense")
    ok = self.gen_idsva_so_mode() == "chain" and not self.gen_idsva_so_direct() and self.tuning["so_mapping"] == "balanced" and self.tuning["so_loops"] == "dots"
    if want == "compact" and not ok:
        raise NotImplementedError("so_stage=compact needs a serial chain whose record is staged in LDS, so_mapping=balanced and so_loops=dots")
    return ok and want != "dense"


def gen_idsva_so_blocks(self):
    """True where the tree form stages a solve's record as the dense BLOCKS of the robot's base-rooted components (tuning so_stage): a fixed base decouples the
    components, so every entry with indices in different components is a structural zero - the quadruped's record shrinks from 6 912 to 432 values (4 x 4 x 3^3),
    the 12-DoF tree's from 6 912 to 4 032; the dense record is gathered through grid_so_expand on the way out, like the compact form of serial chains."""
    want = self.tuning["so_stage"]
    if want not in ("auto", "compact", "dense"):
        raise ValueError("tuning['so_stage'] must be auto, compact or dense")
    return (self.gen_idsva_so_mode() == "tree" and not self.gen_idsva_so_direct() and len(self.model.roots) > 1 and want != "dense" and self.tuning["so_blocked"]
            and self.tuning["so_mapping"] == "balanced" and self.tuning["so_loops"] == "dots")  # (so_blocked: fdsva_so's contraction reads the blocks)


def gen_idsva_so_packed(self):
    """The kernels stage a packed record (compact: serial chains; blocks: trees with several components) and expand it through grid_so_expand."""
    return self.gen_idsva_so_compact() or self.gen_idsva_so_blocks()


def gen_idsva_so_blocks_layout(self):
    """Block staging: component kappa (joints c0 .. c0 + cs - 1) owns 4 cs^3 values at BASE[kappa]: [tensor][i - c0][a - c0][b - c0]; then the ZERO slot.
    Returns SIZE, the per-joint table [c0, cs, base] (flat, one triple per lane of the lane group) and the expansion table."""
    m = self.model
    n = m.n
    comps, base = {}, 0
    for r in m.roots:
        cs = len(m.subtree[r])
        for j_ in m.subtree[r]:
            comps[j_] = (r, cs, base)
        base += 4 * cs ** 3
    L = {"ZERO": base, "SIZE": (base + 1 + 3) // 4 * 4}
    flat = []
    for j_ in range(self.lanes_per_solve):
        flat += list(comps.get(j_, (0, 0, 0)))
    L["JOINTS"] = flat
    table = []
    for ten in range(4):
        for i in range(n):
            for a in range(n):
                for b in range(n):
                    (c0, cs, bs) = comps[i]
                    same = comps[a][0] == c0 and comps[b][0] == c0
                    table.append(bs + ten * cs ** 3 + ((i - c0) * cs + (a - c0)) * cs + (b - c0) if same else L["ZERO"])
    assert max(table) < 65536
    L["TABLE"] = table
    return L


def gen_idsva_so_compact_layout(self):
    if self.gen_idsva_so_blocks():
        return self.gen_idsva_so_blocks_layout()
    return self.gen_idsva_so_chain_compact_layout()


def gen_idsva_so_chain_compact_layout(self):
    """Offsets of the compact staging record and the expansion table (slot of every element of the dense record [d2tau_dq2 | d2tau_dqd2 | d2tau_dvdq | dM_dq]).
    q2c / qd2c: [i][tri(b) + a], a <= b, tri(b) = b (b + 1) / 2;  vq: dense [i][a][b];  mqc: [tri(k) + i][j], i <= k (structurally zero for j <= i:
    those slots are never written and never read - the table sends the elements to the ZERO slot);  then ZERO and a DUMMY slot for predicated-off stores."""
    n = self.model.n
    tri = n * (n + 1) // 2
    off = {"TRI": tri, "Q2": 0, "QD2": n * tri, "VQ": 2 * n * tri, "MQ": 2 * n * tri + n ** 3}
    off["ZERO"] = off["MQ"] + tri * n
    off["DUMMY"] = off["ZERO"] + 1
    off["SIZE"] = (off["DUMMY"] + 1 + 3) // 4 * 4
    t = lambda b: b * (b + 1) // 2
    table = []
    for ten in range(4):
        for i in range(n):
            for a in range(n):
                for b in range(n):
                    if ten < 2:
                        table.append((off["Q2"] if ten == 0 else off["QD2"]) + i * tri + t(max(a, b)) + min(a, b))
                    elif ten == 2:
                        table.append(off["VQ"] + (i * n + a) * n + b)
                    else:  # dM_dq[i][j = a][k = b] = d M_ik / d q_j
                        lo, hi = min(i, b), max(i, b)
                        table.append(off["ZERO"] if a <= lo else off["MQ"] + (t(hi) + lo) * n + a)
    assert max(table) < 65536
    off["TABLE"] = table
    return off


def gen_idsva_so_items_table(self):
    """The items of gen_idsva_so_items as a flat int table [lane][slot][c, m] (-1, -1 = no item), emitted at namespace scope with the model constants
    (a function-local static would be shared between the robot libraries of one process by the host toolchain of the test emulation)."""
    table, slots, _, _ = self.gen_idsva_so_items()
    flat = []
    for ln in range(self.lanes_per_solve):
        for sl in range(slots):
            flat += list(table[ln][sl]) if table[ln][sl] is not None else [-1, -1]
    return flat


def _so_emit_balanced_main(self, tree):
    """Main loops in the balanced mapping (see gen_idsva_so_items); expects the per-lane quantities of _SO_PREP (lane <-> joint) and the records in s_X."""
    n, G = self.model.n, self.lanes_per_solve
    table, slots, tA, tB = self.gen_idsva_so_items()
    A = self.gen_add_code_line
    A("// balanced mapping: this lane's items (c, m), %d per lane, from the table grid_so_items (emitted with the model constants)" % slots)
    A("const bool own_lane = active@NOSTORE@;".replace("@NOSTORE@", " && (gravity < static_cast<T>(-1e30))" if self.tuning["debug_stop"] == 30 else ""))
    A("const T (&oIC)[10] = IC; const T (&oBC)[12] = BC; const T (&oT1)[6] = T1; const T (&oT2)[3] = T2; const T (&oT3)[6] = T3; const T (&oT4)[6] = T4;")
    A("const T (&oICPd)[6] = ICPd; const T (&oS)[6] = S; const T (&oPd)[6] = Pd;")
    par = (lambda l: "static_cast<int>(s_X[20*%s + 18])" % l) if tree else (lambda l: "(%s - 1)" % l)
    for sl in range(slots):
        A("{ // item slot %d: path loops of up to %d steps, ancestor loops of up to %d steps" % (sl, tA[sl], tB[sl]), True)
        A("const int ic = grid_so_items[%d*lane + %d], im = grid_so_items[%d*lane + %d];" % (2 * slots, 2 * sl, 2 * slots, 2 * sl + 1))
        A("const bool item = ic >= 0; const int c = item ? ic : 0, m = item ? im : 0; const bool own = own_lane && item;")
        A("T IC[10], BC[12], T1[6], T2[3], T3[6], T4[6], ICPd[6], S[6], Pd[6]; // joint c's quantities, fetched from its lane")
        for nm, ln_ in (("IC", 10), ("BC", 12), ("T1", 6), ("T2", 3), ("T3", 6), ("T4", 6), ("ICPd", 6), ("S", 6), ("Pd", 6)):
            A("#pragma unroll")
            A("for (int r = 0; r < %d; r++) { %s[r] = grid_group_shfl(o%s[r], c); }" % (ln_, nm, nm))
        _so_emit(self, _SO_OPERATORS)
        def fetch(dst_s, dst_p, dst_par, lv, decl):
            A("%s#pragma unroll" % "")
            A("for (int r = 0; r < 6; r++) { %s[r] = s_X[20*%s + r]; %s[r] = s_X[20*%s + 6 + r]; }" % (dst_s, lv, dst_p, lv))
            A("%s%s = %s;" % ("const int " if decl else "", dst_par, ("static_cast<int>(s_X[20*%s + 18])" % lv) if tree else ("%s - 1" % lv)))

        for (title, start, tmax, more_expr, bodies) in (
                ("joint j = l walks the path c -> m, ancestor-or-self an = m", "c", tA[sl], "va && (l != m)", "A"),
                ("joint j = m, ancestor-or-self an = l walks m -> root", "m", tB[sl], "va && (lpar >= 0)", "BC")):
            A("{ // %s; the record of the next joint is fetched while this one is worked on" % title, True)
            A("int l = %s, lpar; bool va = true;" % start)
            A("T xS[6], xP[6];")
            fetch("xS", "xP", "lpar", "l", False)
            A("#pragma unroll 1")
            A("for (int t = 0; t < %d; t++) {" % tmax, True)
            A("const bool more = %s; const int ln = more ? lpar : l;" % more_expr)
            A("T nS[6], nP[6];")
            fetch("nS", "nP", "npar", "ln", True)
            _so_emit(self, _SO_XDOTS)
            if bodies == "A":
                _so_emit(self, "        {")
                _so_emit(self, _SO_CASE_A, SUB="va")
                _so_emit(self, "        }")
            else:
                _so_emit(self, "        {")
                _so_emit(self, _SO_CASE_B, SUBX="(va && c != j)")
                _so_emit(self, "        }\n        if (l != m) {")
                _so_emit(self, _SO_CASE_C, SUB="va")
                _so_emit(self, "        }")
            A("l = ln; va = more; lpar = npar;")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { xS[r] = nS[r]; xP[r] = nP[r]; }")
            self.gen_add_end_control_flow()
            self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()


# Folded per-item vectors of the dot-product form of the main loops (tuning so_loops = dots).  Every cross product of the reference's p1..p6
# terms pairs a vector of joint l (the loop variable) with vectors of the item (c, m); with (a x b) . f = -b . (a x* f) the item side is folded
# into 6-vectors ONCE per item and a loop step is nothing but dot products with S_l / Pd_l:
#   joint j = l on the path c..m, ancestor-or-self an = m:
#     d2tau_dq2[c][an][j]  = xS.gA - xP.d3P        gA = yP x* [T2;0] - yPP x* T1                        (reference: -p1.T2 + p2.T1 - ...)
#     d2tau_dq2[j][c][an]  = xS.eW                 eW = d2P + d1PP
#     d2tau_dvdq[j][c][an] = xS.eV                 eV = d2S + 2 d1P
#   joint j = m, proper ancestor an = l:
#     d2tau_dvdq[c][j][an] = xS.gC + 2 xP.d4S      gC = d3P - 2 d1P - yS x* [T2;0]     (2 yP x* T1 = 2 (d3P - d1P);  yS x* T1 = d4S)
#     d2tau_dq2[an][j][c]  = xS.e3                 e3 = eW + yS x* T3
#     d2tau_dvdq[an][j][c] = xS.e4                 e4 = d3P + yS x* T4
#     d2tau_dqd2[an][j][j] = 2 xS.d4S   for c = j  (T1.p3 + xS.(S x* T1) with S = yS)
_SO_FOLD = """
    T gA[6], gC[6], eW[6], eV[6], e3[6], e4[6];
    {
        T t[6]; grid_zero6(t); grid_fxv_peq(t, yPP, T1);
        #pragma unroll
        for (int r = 0; r < 6; r++) { gA[r] = -t[r]; gC[r] = d3P[r] - static_cast<T>(2)*d1P[r]; eW[r] = d2P[r] + d1PP[r]; eV[r] = d2S[r] + static_cast<T>(2)*d1P[r]; e3[r] = eW[r]; e4[r] = d3P[r]; }
        gA[0] += yP[1]*T2[2] - yP[2]*T2[1]; gA[1] += yP[2]*T2[0] - yP[0]*T2[2]; gA[2] += yP[0]*T2[1] - yP[1]*T2[0];
        gC[0] -= yS[1]*T2[2] - yS[2]*T2[1]; gC[1] -= yS[2]*T2[0] - yS[0]*T2[2]; gC[2] -= yS[0]*T2[1] - yS[1]*T2[0];
        grid_fxv_peq(e3, yS, T3); grid_fxv_peq(e4, yS, T4);
    }
    const T y_d1S = grid_dot6(yS, d1S); // S_m . D1 S_m: the diagonal d2tau_dqd2[c][m][m] and dM_dq[m][c][m]
"""


class _SoStores:
    """Store statements of the second-order main loops for one staging form.  dense: the record itself, [t][i][a][b] at t n^3 + (i n + a) n + b, symmetric
    entries written twice; compact (gen_idsva_so_compact_layout): every value once.  `tri_b` is the C++ expression of b (b + 1) / 2."""

    def __init__(self, n, compact_layout=None, blocks=False):
        self.n, self.L, self.blocks = n, compact_layout, blocks

    def _blk(self, t, i, a, b, val):  # blocks: the item's component has size cs, cube cs3 and index base cb (= BASE - c0 (cs^2 + cs + 1))
        return "so[cb + %d*cs3 + ((%s)*cs + (%s))*cs + (%s)] = %s;" % (t, i, a, b, val)

    def sym(self, tensor, i, a, b, tri_b, val):  # d2tau_dq2 / d2tau_dqd2 entry (i; a, b) = (i; b, a), a <= b
        n = self.n
        if self.blocks:
            t = 0 if tensor == "q2" else 1
            return self._blk(t, i, a, b, val) + " " + self._blk(t, i, b, a, val)
        if self.L is not None:
            return "so[%d + (%s)*%d + (%s) + (%s)] = %s;" % (self.L["Q2" if tensor == "q2" else "QD2"], i, self.L["TRI"], tri_b, a, val)
        base = 0 if tensor == "q2" else n ** 3
        return "so[%d + ((%s)*%d + (%s))*%d + (%s)] = so[%d + ((%s)*%d + (%s))*%d + (%s)] = %s;" % (base, i, n, a, n, b, base, i, n, b, n, a, val)

    def vq(self, i, a, b, val):
        n = self.n
        if self.blocks:
            return self._blk(2, i, a, b, val)
        return "so[%d + ((%s)*%d + (%s))*%d + (%s)] = %s;" % (self.L["VQ"] if self.L is not None else 2 * n ** 3, i, n, a, n, b, val)

    def mq(self, i, j, k, tri_k, val):  # dM_dq entry [i][j][k] = [k][j][i] = d M_ik / d q_j, i <= k
        n = self.n
        if self.blocks:
            return self._blk(3, i, j, k, val) + " " + self._blk(3, k, j, i, val)
        if self.L is not None:
            return "so[%d + ((%s) + (%s))*%d + (%s)] = %s;" % (self.L["MQ"], tri_k, i, n, j, val)
        return "so[%d + ((%s)*%d + (%s))*%d + (%s)] = so[%d + ((%s)*%d + (%s))*%d + (%s)] = %s;" % (3 * n ** 3, i, n, j, n, k, 3 * n ** 3, k, n, j, n, i, val)


def _so_emit_balanced_main_dots(self, tree, compact, local=False, blocks=False):
    has_pris = any(s_ >= 3 for s_ in self.model.S_index)
    """Main loops, balanced mapping (gen_idsva_so_items), dot-product form (see _SO_FOLD); expects the per-lane quantities of _SO_PREP (lane <-> joint) and
    the records [S | Pd | Pdd | parent] in s_X.  Writes through `so`: the dense record (LDS or global memory) or the compact staging record."""
    n, G = self.model.n, self.lanes_per_solve
    table, slots, tA, tB = self.gen_idsva_so_items()
    st = _SoStores(n, self.gen_idsva_so_chain_compact_layout() if compact else None, blocks=blocks)
    A = self.gen_add_code_line
    A("// balanced mapping: this lane's items (c, m), %d per lane, from the table grid_so_items (emitted with the model constants)" % slots)
    A("const bool own_lane = active@NOSTORE@;".replace("@NOSTORE@", " && (gravity < static_cast<T>(-1e30))" if self.tuning["debug_stop"] == 30 else ""))
    A("const T (&oIC)[10] = IC; const T (&oBC)[12] = BC; const T (&oT1)[6] = T1; const T (&oT2)[3] = T2; const T (&oT3)[6] = T3; const T (&oT4)[6] = T4;")
    A("const T (&oICPd)[6] = ICPd; const T (&oS)[6] = S; const T (&oPd)[6] = Pd;")
    par = (lambda l: "static_cast<int>(s_X[20*%s + 18])" % l) if tree else (lambda l: "(%s - 1)" % l)
    tri = lambda v: "((%s)*((%s) + 1) >> 1)" % (v, v)
    for sl in range(slots):
        A("{ // item slot %d: path loops of up to %d steps, ancestor loops of up to %d steps" % (sl, tA[sl], tB[sl]), True)
        A("const int ic = grid_so_items[%d*lane + %d], im = grid_so_items[%d*lane + %d];" % (2 * slots, 2 * sl, 2 * slots, 2 * sl + 1))
        A("const bool item = ic >= 0; const int c = item ? ic : 0, m = item ? im : 0; const bool own = own_lane && item;")
        A("const int tc = %s, tm = %s; (void)tc; (void)tm; // c (c + 1) / 2, m (m + 1) / 2: rows of the symmetric (compact) index" % (tri("c"), tri("m")))
        if blocks:
            A("const int cs = grid_so_blocks[3*c + 1], cs3 = cs*cs*cs, cb = grid_so_blocks[3*c + 2] - grid_so_blocks[3*c]*(cs*cs + cs + 1); // block of joint c's base-rooted component (every joint of the item lies in it)")
        A("T IC[10], BC[12], T1[6], T2[3], T3[6], T4[6], ICPd[6], S[6], Pd[6]; // joint c's quantities, fetched from its lane")
        for nm, ln_ in (("IC", 10), ("BC", 12), ("T1", 6), ("T2", 3), ("T3", 6), ("T4", 6), ("ICPd", 6), ("S", 6), ("Pd", 6)):
            A("#pragma unroll")
            A("for (int r = 0; r < %d; r++) { %s[r] = grid_group_shfl(o%s[r], c); }" % (ln_, nm, nm))
        if local:
            # records hold every joint's vectors about its OWN origin ([axis | origin | Pd | Pdd]): moved to joint c's origin before they meet its composites
            A("T pc[3] = {s_X[20*c + 3], s_X[20*c + 4], s_X[20*c + 5]}; // origin of joint c's frame: the reference point of this item")
            _so_emit(self, _SO_OPERATORS.replace("""    T yS[6], yP[6], yPP[6];
    #pragma unroll
    for (int r = 0; r < 6; r++) { yS[r] = s_X[20*m + r]; yP[r] = s_X[20*m + 6 + r]; yPP[r] = s_X[20*m + 12 + r]; }""",
                                                 """    T yS[6], yP[6], yPP[6];
    { const T *ry = &s_X[20*m]; const T qm[3] = {pc[0] - ry[3], pc[1] - ry[4], pc[2] - ry[5]};
      @YAXIS@ grid_so_motion_at(yP, ry + 6, qm); grid_so_motion_at(yPP, ry + 12, qm); }""".replace("@YAXIS@", "grid_so_joint_axis_at(yS, ry, qm, ry[19] != static_cast<T>(0));" if has_pris else "grid_so_axis_at(yS, ry, qm);")))
        else:
            _so_emit(self, _SO_OPERATORS)
        _so_emit(self, _SO_FOLD)

        def fetch(dst_s, dst_p, dst_par, lv, decl):
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { %s[r] = s_X[20*%s + r]; %s[r] = s_X[20*%s + 6 + r]; }" % (dst_s, lv, dst_p, lv))
            A("%s%s = %s;" % ("const int " if decl else "", dst_par, par(lv)))
            if local and has_pris:
                A("%s%s = s_X[20*%s + 19]; // joint type" % ("const T " if decl else "", dst_par.replace("par", "typ"), lv))

        def xvec():  # the step's vectors S_l, Pd_l from the fetched record words
            if local:
                A("T xS[6], xP[6];")
                A("{ const T ql[3] = {pc[0] - rS[3], pc[1] - rS[4], pc[2] - rS[5]}; %s grid_so_motion_at(xP, rP, ql); }" %
                  ("grid_so_joint_axis_at(xS, rS, ql, ltyp != static_cast<T>(0));" if has_pris else "grid_so_axis_at(xS, rS, ql);"))
            else:
                A("const T (&xS)[6] = rS; const T (&xP)[6] = rP;")

        A("if (own && c != m) { %s } // dM_dq[m][c][m]: the step l = m of the ancestor loop" % st.mq("m", "c", "m", "tm", "y_d1S"))
        # ---- path loop: joint j = l walks c -> m, ancestor-or-self an = m
        A("{ // joint j = l walks the path c -> m, ancestor-or-self an = m; the record of the next joint is fetched while this one is worked on", True)
        A("int l = c, lpar; bool va = true;" + (" T ltyp;" if (local and has_pris) else ""))
        A("T rS[6], rP[6];")
        fetch("rS", "rP", "lpar", "l", False)
        A("#pragma unroll 1")
        A("for (int t = 0; t < %d; t++) {" % tA[sl], True)
        A("const bool more = va && (l != m); const int ln = more ? lpar : l;")
        A("T nS[6], nP[6];")
        fetch("nS", "nP", "npar", "ln", True)
        xvec()
        A("const T s_d3P = grid_dot6(xS, d3P), s_d3S = grid_dot6(xS, d3S), s_eW = grid_dot6(xS, eW), s_eV = grid_dot6(xS, eV);")
        A("const T vq_ = grid_dot6(xS, gA) - grid_dot6(xP, d3P);")
        A("const int j = l, an = m, tj = %s; (void)tj;" % tri("l"))
        A("const bool ok = own && va, ne = ok && (c != j);")
        A("if (ok) {", True)
        A(st.sym("q2", "c", "an", "j", "tj", "vq_"))
        A(st.vq("c", "an", "j", "-s_d3P"))
        A(st.sym("qd2", "c", "an", "j", "tj", "(an != j) ? -s_d3S : -y_d1S"))
        self.gen_add_end_control_flow()
        A("if (ne) {", True)
        A(st.sym("q2", "j", "an", "c", "tc", "s_eW"))
        A(st.vq("j", "an", "c", "s_d3P"))
        A(st.sym("qd2", "j", "an", "c", "tc", "s_d3S"))
        A(st.vq("j", "c", "an", "s_eV"))
        self.gen_add_end_control_flow()
        A("l = ln; va = more; lpar = npar;" + (" ltyp = ntyp;" if (local and has_pris) else ""))
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { rS[r] = nS[r]; rP[r] = nP[r]; }")
        self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
        # ---- ancestor loop: joint j = m, proper ancestor an = l walks parent(m) -> root
        if tB[sl] > 0:
            A("{ // joint j = m, proper ancestor an = l walks parent(m) -> root", True)
            A("int l = %s, lpar; bool va = l >= 0; if (!va) { l = 0; }" % par("m") + (" T ltyp;" if (local and has_pris) else ""))
            A("T rS[6], rP[6];")
            fetch("rS", "rP", "lpar", "l", False)
            A("#pragma unroll 1")
            A("for (int t = 0; t < %d; t++) {" % tB[sl], True)
            A("const bool more = va && (lpar >= 0); const int ln = more ? lpar : l;")
            A("T nS[6], nP[6];")
            fetch("nS", "nP", "npar", "ln", True)
            xvec()
            A("const T s_d1S = grid_dot6(xS, d1S), s_d3S = grid_dot6(xS, d3S), s_d4S = grid_dot6(xS, d4S), s_eV = grid_dot6(xS, eV), s_e3 = grid_dot6(xS, e3), s_e4 = grid_dot6(xS, e4);")
            A("const T vq_ = grid_dot6(xS, gC) + static_cast<T>(2)*grid_dot6(xP, d4S);")
            A("const int j = m, an = l;")
            A("const bool ok = own && va, ne = ok && (c != j);")
            A("if (ok) {", True)
            A(st.vq("c", "j", "an", "vq_"))
            A(st.sym("q2", "an", "j", "c", "tc", "s_e3"))
            A(st.vq("an", "j", "c", "s_e4"))
            A(st.mq("an", "j", "c", "tc", "s_d4S"))
            A(st.sym("qd2", "an", "j", "c", "tc", "(c != j) ? s_d3S : static_cast<T>(2)*s_d4S"))
            self.gen_add_end_control_flow()
            A("if (ne) {", True)
            A(st.vq("an", "c", "j", "s_eV"))
            A(st.mq("an", "c", "j", "tm", "s_d1S"))
            self.gen_add_end_control_flow()
            A("l = ln; va = more; lpar = npar;" + (" ltyp = ntyp;" if (local and has_pris) else ""))
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { rS[r] = nS[r]; rP[r] = nP[r]; }")
            self.gen_add_end_control_flow()
            self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()


_SO_PREP = """
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
    @PARENT@
}
grid_wave_sync();
const bool own = active && (lane < @N@)@NOSTORE@;
T *q2 = so, *qd2 = so + @N3@, *vq = so + 2*@N3@, *mq = so + 3*@N3@;
const int c = lane;
"""

# operators of lane c applied to the vectors of joint m (identical for chains and trees)
_SO_OPERATORS = """
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
"""

_SO_XDOTS = """
        const T x_d1S = grid_dot6(xS, d1S), x_d3S = grid_dot6(xS, d3S), x_d2S = grid_dot6(xS, d2S), x_d4S = grid_dot6(xS, d4S);
        const T x_d1P = grid_dot6(xS, d1P), x_d3P = grid_dot6(xS, d3P), x_d2P = grid_dot6(xS, d2P), x_d1PP = grid_dot6(xS, d1PP);
"""
_SO_X = """
        T xS[6], xP[6];
        #pragma unroll
        for (int r = 0; r < 6; r++) { xS[r] = s_X[20*l + r]; xP[r] = s_X[20*l + 6 + r]; }
""" + _SO_XDOTS

# joint j = l, ancestor-or-self an = m (reference phases t1..t5 and the p1/p2 terms, :566-710, :876-886); @SUB@ = "c is in the subtree of j (incl. j)"
_SO_CASE_A = """
            const int j = l, an = m;
            const bool ok = own && @SUB@, ne = ok && (c != j);
            T p1[6], p2[6]; grid_mxm(p1, yP, xS); grid_mxm(p2, yPP, xS);
            const T pa = -(p1[0]*T2[0] + p1[1]*T2[1] + p1[2]*T2[2]) + grid_dot6(p2, T1);
            const T vq_ = -grid_dot6(xP, d3P) + pa;
            const T w_ = x_d2P + x_d1PP;
            if (ok) {
                q2[(c*@N@ + an)*@N@ + j] = vq_;
                vq[(c*@N@ + an)*@N@ + j] = -x_d3P;
                if (an != j) { q2[(c*@N@ + j)*@N@ + an] = vq_; qd2[(c*@N@ + j)*@N@ + an] = -x_d3S; qd2[(c*@N@ + an)*@N@ + j] = -x_d3S; }
                else { qd2[(c*@N@ + an)*@N@ + j] = -x_d1S; }
            }
            if (ne) {
                q2[(j*@N@ + c)*@N@ + an] = w_; q2[(j*@N@ + an)*@N@ + c] = w_;
                vq[(j*@N@ + an)*@N@ + c] = x_d3P;
                qd2[(j*@N@ + c)*@N@ + an] = x_d3S; qd2[(j*@N@ + an)*@N@ + c] = x_d3S;
                vq[(j*@N@ + c)*@N@ + an] = x_d2S + static_cast<T>(2)*x_d1P;
            }
"""

# joint j = m, ancestor-or-self an = l: second half of the reference's t8 phase (:815-816); @SUBX@ = "c is in the subtree of j, c != j"
_SO_CASE_B = """
            const int j = m, an = l;
            if (own && @SUBX@) { mq[(an*@N@ + c)*@N@ + j] = x_d1S; mq[(j*@N@ + c)*@N@ + an] = x_d1S; }
"""

# joint j = m, proper ancestor an = l (reference phases t6..t9, the p3..p6 terms, :725-911)
_SO_CASE_C = """
            const int j = m, an = l;
            const bool ok = own && @SUB@, ne = ok && (c != j);
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
"""


def _so_emit(self, text, **subs):
    n = self.model.n
    text = text.replace("@N3@", str(n * n * n)).replace("@N@", str(n))
    text = text.replace("@NOSTORE@", " && (gravity < static_cast<T>(-1e30))" if self.tuning["debug_stop"] == 30 else "")  # timing ablation: everything computed, nothing stored
    for k, v in subs.items():
        text = text.replace("@" + k + "@", v)
    for line in text.strip("\n").split("\n"):
        self.gen_add_code_line(line)


def _so_inner_header(self, compact=False):
    n = self.model.n
    form = "serial revolute chains; computed in the frame of the tip link" if self.gen_idsva_so_mode() == "chain" else \
        "kinematic trees of revolute joints; computed in the base frame, kinematics and subtree composites handed level by level through LDS"
    self.gen_add_func_doc("Computes the second order derivatives of inverse dynamics",
                          [form + ", lane c owns the subtree joint of every (joint, ancestor, subtree) triple",
                           "so = [d2tau_dq2 | d2tau_dqd2 | d2tau_dvdq | dM_dq], each n x n x n with [i][j][k] at i*n*n + j*n + k (reference algorithms/_idsva_so.py:204-208);",
                           "d2tau_dvdq[i][j][k] = d2 tau_i / dq_j dqd_k, dM_dq[i][j][k] = d M_ik / dq_j.  Every entry is written exactly once (trees: after a zero fill of the record by the same wave); `so` may be LDS" +
                           (" or global memory" if self.gen_idsva_so_mode() == "chain" else "")] +
                          (["COMPACT form: `so` is the staging record of IDSVA_SO_STAGE_PER_SOLVE values (serial chains: every value once - symmetric entries, no structural zeros; trees: the dense blocks of the base-rooted components); the dense",
                            "record is gathered from it through the table grid_so_expand (what the kernels do when the record leaves for global memory)"] if compact else []),
                          [("so is the compact staging record of this solve (IDSVA_SO_STAGE_PER_SOLVE values)" if compact else "so is the output record of this solve (4*NUM_JOINTS^3 values)"), "s_qd is the vector of joint velocities in LDS",
                           "s_qdd is the vector of joint accelerations in LDS", "s_X is this solve's scratch: compact X(q) storage on entry; it is overwritten by the per-joint records",
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU", "gravity is the gravity constant",
                           "lane is the caller's lane index inside the solve's lane group", "active is false for lane groups without a solve (they compute but do not store)"] +
                          (["blocks_only (robots with several base-rooted components): zero-fill only the entries whose three indices lie in ONE component - the others are never written"] if (self.gen_idsva_so_mode() == "tree" and len(self.model.roots) > 1) else []), None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    zb = ", const bool blocks_only = false" if (self.gen_idsva_so_mode() == "tree" and len(self.model.roots) > 1) else ""
    self.gen_add_code_line("void idsva_so_inner" + ("_compact" if compact else "") + "(T *__restrict__ so, const T *__restrict__ s_qd, const T *__restrict__ s_qdd, T *__restrict__ s_X, const robotModel<T> *d_robotModel, const T gravity, const int lane, const bool active" + zb + ") {", True)


def gen_idsva_so_inner_chain(self, use_thread_group=False, compact=False):
    n = self.model.n
    _so_inner_header(self, compact)
    _emit_link_constants_load(self)
    _emit_chain_decls(self)
    for i in range(self.tip_L - 1, -1, -1):
        _chain_step(self, i)
    self.gen_add_sync(use_thread_group)  # every lane is done with s_X before the records overwrite it
    _emit_link_setup(self)
    self.gen_add_code_line("const T qdd = (lane < %d) ? s_qdd[lane] : static_cast<T>(0);" % n)
    _emit_bias(self, True)
    _so_emit(self, _SO_PREP, PARENT="")
    if compact:
        L = self.gen_idsva_so_compact_layout()
        self.gen_add_code_line("if (lane == 0) { so[%d] = static_cast<T>(0); } // the ZERO slot: what the structurally zero entries of dM_dq expand from" % L["ZERO"])
    else:
        _so_emit(self, """
// structurally zero entries of dM_dq: dM_ik/dq_j with j <= min(i, k); the owner lane (largest index) writes them first
#pragma unroll 1
for (int b = 0; b < @N@; b++) {
    for (int e = 0; e <= b; e++) {
        if (own && b <= c) { mq[(c*@N@ + e)*@N@ + b] = static_cast<T>(0); mq[(b*@N@ + e)*@N@ + c] = static_cast<T>(0); }
    }
}
""")
    if self.tuning["so_mapping"] == "balanced":
        if self.tuning["so_loops"] == "dots":
            _so_emit_balanced_main_dots(self, tree=False, compact=compact)
        else:
            _so_emit_balanced_main(self, tree=False)
        self.gen_add_end_function()
        return
    _so_emit(self, """
// main loops: joint m supplies the vectors this lane's operators act on, joint l the vector the results are dotted with
#pragma unroll 1
for (int m = 0; m < @N@; m++) {""")
    _so_emit(self, _SO_OPERATORS)
    _so_emit(self, "    #pragma unroll @UNROLL_L@\n    for (int l = 0; l < @N@; l++) {", UNROLL_L=str(self.tuning["so_unroll"] or n))
    _so_emit(self, _SO_X)
    _so_emit(self, "        if (l >= m) {")
    _so_emit(self, _SO_CASE_A, SUB="(c >= j)")
    _so_emit(self, "        }\n        if (l <= m) {")
    _so_emit(self, _SO_CASE_B, SUBX="(c > j)")
    _so_emit(self, "        }\n        if (l < m) {")
    _so_emit(self, _SO_CASE_C, SUB="(c >= j)")
    _so_emit(self, "        }\n    }\n}")
    self.gen_add_end_function()


_SO_SHIFT_LIBRARY = """
// --- change of reference point (same axes): q = position of the NEW origin relative to the old one -------------------------------------
// motion vector [w; u] (u = velocity of the point at the origin): u' = u + w x q
template <typename T>
__device__ __forceinline__ void grid_so_motion_at(T (&y)[6], const T *x, const T (&q)[3]) {
    y[0] = x[0]; y[1] = x[1]; y[2] = x[2];
    y[3] = x[3] + x[1]*q[2] - x[2]*q[1]; y[4] = x[4] + x[2]*q[0] - x[0]*q[2]; y[5] = x[5] + x[0]*q[1] - x[1]*q[0];
}
// joint axis through the old origin (linear part zero there): [w; w x q]
template <typename T>
__device__ __forceinline__ void grid_so_axis_at(T (&y)[6], const T *w, const T (&q)[3]) {
    y[0] = w[0]; y[1] = w[1]; y[2] = w[2];
    y[3] = w[1]*q[2] - w[2]*q[1]; y[4] = w[2]*q[0] - w[0]*q[2]; y[5] = w[0]*q[1] - w[1]*q[0];
}
// motion subspace of a joint whose frame origin is the old reference point: revolute [w; w x q], prismatic [0; w] (a free vector)
template <typename T>
__device__ __forceinline__ void grid_so_joint_axis_at(T (&y)[6], const T *w, const T (&q)[3], const bool prismatic) {
    const T z = static_cast<T>(0);
    y[0] = prismatic ? z : w[0]; y[1] = prismatic ? z : w[1]; y[2] = prismatic ? z : w[2];
    y[3] = prismatic ? w[0] : (w[1]*q[2] - w[2]*q[1]); y[4] = prismatic ? w[1] : (w[2]*q[0] - w[0]*q[2]); y[5] = prismatic ? w[2] : (w[0]*q[1] - w[1]*q[0]);
}
// y += [I^C (10) | B^C (12) | f^C (6)] of a subtree, moved to the new reference point (rigid-body inertia: h' = h - m q, parallel axes;
// Coriolis matrix [Sym | n | l]: Sym' = Sym + l q^T + q l^T - 2 (q.l) 1, n' = n - q x l; force [n; f]: n' = n - q x f)
template <typename T>
__device__ __forceinline__ void grid_so_composite_peq(T (&IC)[10], T (&BC)[12], T (&fC)[6], const T *rec, const T (&q)[3]) {
    {
        const T *I = rec; const T m = I[9];
        const T s = static_cast<T>(-2)*(I[6]*q[0] + I[7]*q[1] + I[8]*q[2]) + m*(q[0]*q[0] + q[1]*q[1] + q[2]*q[2]);
        IC[0] += I[0] + s + static_cast<T>(2)*I[6]*q[0] - m*q[0]*q[0];
        IC[1] += I[1] + I[6]*q[1] + q[0]*I[7] - m*q[0]*q[1];
        IC[2] += I[2] + I[6]*q[2] + q[0]*I[8] - m*q[0]*q[2];
        IC[3] += I[3] + s + static_cast<T>(2)*I[7]*q[1] - m*q[1]*q[1];
        IC[4] += I[4] + I[7]*q[2] + q[1]*I[8] - m*q[1]*q[2];
        IC[5] += I[5] + s + static_cast<T>(2)*I[8]*q[2] - m*q[2]*q[2];
        IC[6] += I[6] - m*q[0]; IC[7] += I[7] - m*q[1]; IC[8] += I[8] - m*q[2]; IC[9] += m;
    }
    {
        const T *B = rec + 10; const T l0 = B[9], l1 = B[10], l2 = B[11];
        const T s = static_cast<T>(-2)*(q[0]*l0 + q[1]*l1 + q[2]*l2);
        BC[0] += B[0] + s + static_cast<T>(2)*l0*q[0];
        BC[1] += B[1] + l0*q[1] + q[0]*l1;
        BC[2] += B[2] + l0*q[2] + q[0]*l2;
        BC[3] += B[3] + s + static_cast<T>(2)*l1*q[1];
        BC[4] += B[4] + l1*q[2] + q[1]*l2;
        BC[5] += B[5] + s + static_cast<T>(2)*l2*q[2];
        BC[6] += B[6] - (q[1]*l2 - q[2]*l1); BC[7] += B[7] - (q[2]*l0 - q[0]*l2); BC[8] += B[8] - (q[0]*l1 - q[1]*l0);
        BC[9] += l0; BC[10] += l1; BC[11] += l2;
    }
    {
        const T *f = rec + 22;
        fC[0] += f[0] - (q[1]*f[5] - q[2]*f[4]); fC[1] += f[1] - (q[2]*f[3] - q[0]*f[5]); fC[2] += f[2] - (q[0]*f[4] - q[1]*f[3]);
        fC[3] += f[3]; fC[4] += f[4]; fC[5] += f[5];
    }
}
"""


def gen_idsva_so_local_origin(self):
    """True where the tree form keeps every joint's quantities about the origin of its own frame (tuning so_origin)."""
    return self.gen_idsva_so_mode() == "tree" and self.tuning["so_origin"] == "joint" and self.tuning["so_mapping"] == "balanced" and self.tuning["so_loops"] == "dots"


def gen_idsva_so_inner_tree(self, use_thread_group=False, blocks=False):
    """Tree form.  Same operators and the same merged per-entry formulas as the chain form; what changes is where the per-joint quantities come from
    (level-by-level propagation through LDS instead of lane scans, all in the base frame) and which (m, l) pairs are visited: the reference's triples
    (joint j, ancestor-or-self an, subtree member c) live on ONE root path, so for every m only its subtree (a contiguous id range in DFS pre-order)
    and its ancestors (a parent walk) are visited - unrelated joints contribute structural zeros, which a zero fill of the record provides."""
    m_ = self.model
    n = m_.n
    fl, it, K, maxc = self.gen_idsva_so_tree_tables()
    depth = max(m_.depth)
    off = self.so_tree_tab_offset
    local = gen_idsva_so_local_origin(self)
    if local and not getattr(self, "_so_shift_lib_done", False):
        for line in _SO_SHIFT_LIBRARY.strip("\n").split("\n"):
            self.gen_add_code_line(line)
        self.gen_add_code_line("")
        self._so_shift_lib_done = True
    _so_inner_header(self, compact=blocks)
    A = self.gen_add_code_line
    A("const int *tp = &grid_so_tree_topology[%d*lane]; // this lane's row: parent, level, subtree size, number of children, children" % K)
    A("const int par = tp[0], lev = tp[1];")
    A("T Lc[16]; // link constants: Ic (6, about the centre of mass), c (3), m, damping, axis | origin of the joint frame in its parent's coordinates (3)")
    A("{ const T *d_L = &grid_model_constants(static_cast<const T *>(nullptr))[%d + 16*lane]; (void)d_robotModel;" % off)
    A("  #pragma unroll")
    A("  for (int r = 0; r < 16; r++) { Lc[r] = d_L[r]; } }")
    A("const bool has = lane < %d;" % n)
    A("T E[9]; // rotation block of this joint's X(q) (parent -> child coordinates), fetched before the scratch is re-used")
    A("#pragma unroll")
    A("for (int r = 0; r < 9; r++) { E[r] = has ? s_X[GRID_X_STRIDE*lane + r] : static_cast<T>(0); }")
    A("const T qd = has ? s_qd[lane] : static_cast<T>(0), qdd = has ? s_qdd[lane] : static_cast<T>(0);")
    has_pris = any(s_ >= 3 for s_ in m_.S_index)
    if has_pris:
        A("const bool pris = static_cast<int>(Lc[11]) >= 3; // prismatic joint: the origin of its frame moves with q - read it off X(q) = [[E, 0], [-E r~, E]]: r~ = -E^T B")
        A("{ T Bq[9];")
        A("  #pragma unroll")
        A("  for (int r = 0; r < 9; r++) { Bq[r] = has ? s_X[GRID_X_STRIDE*lane + 9 + r] : static_cast<T>(0); }")
        A("  const T r0 = -(E[2]*Bq[1] + E[5]*Bq[4] + E[8]*Bq[7]), r1 = -(E[0]*Bq[2] + E[3]*Bq[5] + E[6]*Bq[8]), r2 = -(E[1]*Bq[0] + E[4]*Bq[3] + E[7]*Bq[6]);")
        A("  if (pris) { Lc[12] = r0; Lc[13] = r1; Lc[14] = r2; } }")
    self.gen_add_sync(use_thread_group)
    A("// zero fill of the output record: most of its 4 n^3 entries are structural zeros of the tree (joints on different root paths)")
    if blocks:
        A("if (active) { for (int e = lane; e < %d; e += GRID_LANES_PER_SOLVE) { so[e] = static_cast<T>(0); } } // (the component blocks and the ZERO slot that everything else expands from)" % self.gen_idsva_so_blocks_layout()["SIZE"])
    elif len(m_.roots) > 1:
        A("if (active && blocks_only) { // (what fdsva_so_contract_kernel reads: the blocks of the base-rooted components; rows of `size` contiguous entries)", True)
        for r_ in m_.roots:
            c0, cs = r_, len(m_.subtree[r_])
            A("for (int r = lane; r < %d; r += GRID_LANES_PER_SOLVE) { const int t = r / %d, ia = r %% %d; T *row = &so[t*%d + ((%d + ia / %d)*%d + %d + ia %% %d)*%d + %d];" % (4 * cs * cs, cs * cs, cs * cs, n ** 3, c0, cs, n, c0, cs, n, c0))
            A("  #pragma unroll")
            A("  for (int b = 0; b < %d; b++) { row[b] = static_cast<T>(0); } }" % cs)
        self.gen_add_end_control_flow()
        A("else if (active) { for (int e = lane; e < %d; e += GRID_LANES_PER_SOLVE) { so[e] = static_cast<T>(0); } }" % (4 * n ** 3))
    else:
        A("if (active) { for (int e = lane; e < %d; e += GRID_LANES_PER_SOLVE) { so[e] = static_cast<T>(0); } }" % (4 * n ** 3))
    A("// top-down, one tree level at a time: pose of the link frame in the base frame (myR: link -> base coordinates, myp: its origin), joint axis S,")
    A("// spatial velocity v, acceleration a (with the gravity term) and Pd = v_parent x S; a joint's record [myR | myp | v | a] waits in LDS for its children")
    A("T myR[9], myp[3], S[6], v[6], a[6], Pd[6], dpar[3] = {static_cast<T>(0), static_cast<T>(0), static_cast<T>(0)}; // (dpar: origin of this joint's frame relative to its parent's, base axes)")
    if has_pris:
        A("T wax[3] = {static_cast<T>(0), static_cast<T>(0), static_cast<T>(0)};")
    A("#pragma unroll")
    A("for (int r = 0; r < 6; r++) { S[r] = v[r] = a[r] = Pd[r] = static_cast<T>(0); }")
    A("#pragma unroll")
    A("for (int r = 0; r < 9; r++) { myR[r] = static_cast<T>(0); }")
    A("myp[0] = myp[1] = myp[2] = static_cast<T>(0);")
    A("#pragma unroll 1")
    A("for (int L = 0; L <= %d; L++) {" % depth, True)
    A("if (has && lev == L) {", True)
    A("T Rp[9] = {static_cast<T>(1), static_cast<T>(0), static_cast<T>(0), static_cast<T>(0), static_cast<T>(1), static_cast<T>(0), static_cast<T>(0), static_cast<T>(0), static_cast<T>(1)};")
    A("T pp[3] = {static_cast<T>(0), static_cast<T>(0), static_cast<T>(0)}, vp[6], ap[6];")
    A("#pragma unroll")
    A("for (int r = 0; r < 6; r++) { vp[r] = static_cast<T>(0); ap[r] = static_cast<T>(0); }")
    A("ap[5] = gravity; // base acceleration (0, 0, g): gravity is passed positive (reference algorithms/_inverse_dynamics.py:123)")
    A("if (par >= 0) {", True)
    A("const T *rec = &s_X[28*par];")
    A("#pragma unroll")
    A("for (int r = 0; r < 9; r++) { Rp[r] = rec[r]; }")
    A("#pragma unroll")
    A("for (int r = 0; r < 3; r++) { pp[r] = rec[9 + r]; }")
    A("#pragma unroll")
    A("for (int r = 0; r < 6; r++) { vp[r] = rec[12 + r]; ap[r] = rec[18 + r]; }")
    self.gen_add_end_control_flow()
    A("#pragma unroll")
    A("for (int r = 0; r < 3; r++) { // myR = Rp E^T, myp = pp + Rp r_joint", True)
    A("#pragma unroll")
    A("for (int cc = 0; cc < 3; cc++) { myR[3*r + cc] = Rp[3*r]*E[3*cc] + Rp[3*r+1]*E[3*cc+1] + Rp[3*r+2]*E[3*cc+2]; }")
    A("dpar[r] = Rp[3*r]*Lc[12] + Rp[3*r+1]*Lc[13] + Rp[3*r+2]*Lc[14]; myp[r] = pp[r] + dpar[r];")
    self.gen_add_end_control_flow()
    A("{ const int ax = static_cast<int>(Lc[11]) % 3;" if has_pris else "{ const int ax = static_cast<int>(Lc[11]);")
    A("  #pragma unroll")
    A("  for (int r = 0; r < 3; r++) { S[r] = (ax == 0) ? myR[3*r] : ((ax == 1) ? myR[3*r+1] : myR[3*r+2]); } }")
    lin = ("S[3] = S[4] = S[5] = static_cast<T>(0);" if local else
           "S[3] = myp[1]*S[2] - myp[2]*S[1]; S[4] = myp[2]*S[0] - myp[0]*S[2]; S[5] = myp[0]*S[1] - myp[1]*S[0];")
    if has_pris:
        A("wax[0] = S[0]; wax[1] = S[1]; wax[2] = S[2]; // the joint axis in base coordinates")
        A("if (pris) { S[3] = S[0]; S[4] = S[1]; S[5] = S[2]; S[0] = S[1] = S[2] = static_cast<T>(0); } // prismatic: [0; axis], a free vector")
        A("else { %s }" % lin)
    else:
        A(lin)
    if local:
        A("// (local origins: the parent's velocity and acceleration arrive about ITS origin and move to this joint's - the kinematics never see the far-away base origin)")
        A("{ T t[6]; grid_so_motion_at(t, vp, dpar); vp[3] = t[3]; vp[4] = t[4]; vp[5] = t[5]; grid_so_motion_at(t, ap, dpar); ap[3] = t[3]; ap[4] = t[4]; ap[5] = t[5]; }")
    A("grid_mxm(Pd, vp, S);")
    A("#pragma unroll")
    A("for (int r = 0; r < 6; r++) { v[r] = vp[r] + S[r]*qd; a[r] = ap[r] + Pd[r]*qd + S[r]*qdd; }")
    A("T *out = &s_X[28*lane];")
    A("#pragma unroll")
    A("for (int r = 0; r < 9; r++) { out[r] = myR[r]; }")
    A("#pragma unroll")
    A("for (int r = 0; r < 3; r++) { out[9 + r] = myp[r]; }")
    A("#pragma unroll")
    A("for (int r = 0; r < 6; r++) { out[12 + r] = v[r]; out[18 + r] = a[r]; }")
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)
    self.gen_add_end_control_flow()
    if local:
        A("// from here on every quantity of this joint is taken about the ORIGIN OF ITS OWN FRAME (base axes): the joint axis passes through it, the link's inertia, Coriolis")
        A("// matrix and force and their subtree composites are those of a body next to the reference point - about the base origin the entries of light distal links")
        A("// are small differences of m d^2 terms, and fdsva_so multiplies them by M^-1 twice (fp32: 4e-7 of max|dM_dq| became 8e-4 of max|d2a_dtdq| on the 12-DoF tree)")
        A("// (S, v, a and Pd = v_parent x S were propagated about the joints' own origins already)")
    _emit_link_inertia(self, local_origin=local)
    _emit_body_terms(self)
    A("#pragma unroll")
    A("for (int r = 0; r < 10; r++) { IC[r] = has ? I[r] : static_cast<T>(0); }")
    A("// bottom-up, one tree level at a time: composites over the subtree, [I^C | B^C | f^C] of a joint waits in LDS for its parent" + (" (about the joint's own origin; the parent moves it to its own)" if local else ""))
    A("#pragma unroll 1")
    A("for (int L = %d; L >= 0; L--) {" % depth, True)
    A("if (has && lev == L) {", True)
    A("#pragma unroll 1")
    A("for (int ci = 0; ci < tp[3]; ci++) {", True)
    A("const T *rec = &s_X[28*tp[4 + ci]]; // (local origins: the child has moved its composite to THIS joint's origin)")
    A("#pragma unroll")
    A("for (int r = 0; r < 10; r++) { IC[r] += rec[r]; }")
    A("#pragma unroll")
    A("for (int r = 0; r < 12; r++) { BC[r] += rec[10 + r]; }")
    A("#pragma unroll")
    A("for (int r = 0; r < 6; r++) { fC[r] += rec[22 + r]; }")
    self.gen_add_end_control_flow()
    A("T *out = &s_X[28*lane];")
    if local:
        A("{ // what the parent adds: this subtree's composite about the PARENT's origin (-dpar from here)", True)
        A("T own[28], Iu[10], Bu[12], fu[6]; const T q[3] = {-dpar[0], -dpar[1], -dpar[2]};")
        A("#pragma unroll")
        A("for (int r = 0; r < 10; r++) { own[r] = IC[r]; Iu[r] = static_cast<T>(0); }")
        A("#pragma unroll")
        A("for (int r = 0; r < 12; r++) { own[10 + r] = BC[r]; Bu[r] = static_cast<T>(0); }")
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { own[22 + r] = fC[r]; fu[r] = static_cast<T>(0); }")
        A("grid_so_composite_peq(Iu, Bu, fu, own, q);")
        A("#pragma unroll")
        A("for (int r = 0; r < 10; r++) { out[r] = Iu[r]; }")
        A("#pragma unroll")
        A("for (int r = 0; r < 12; r++) { out[10 + r] = Bu[r]; }")
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { out[22 + r] = fu[r]; }")
        self.gen_add_end_control_flow()
    else:
        A("#pragma unroll")
        A("for (int r = 0; r < 10; r++) { out[r] = IC[r]; }")
        A("#pragma unroll")
        A("for (int r = 0; r < 12; r++) { out[10 + r] = BC[r]; }")
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { out[22 + r] = fC[r]; }")
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)
    self.gen_add_end_control_flow()
    _so_emit(self, _SO_PREP, PARENT="rec[18] = static_cast<T>(par); // parent joint id (-1: base), read by the balanced main loops" +
             ("\n    rec[3] = myp[0]; rec[4] = myp[1]; rec[5] = myp[2]; // (the linear part of S is zero about the joint's own origin: its slots carry the origin)" if local else "") +
             ("\n    rec[0] = wax[0]; rec[1] = wax[1]; rec[2] = wax[2]; rec[19] = pris ? static_cast<T>(1) : static_cast<T>(0); // (axis and joint type: a prismatic joint's S is [0; axis])" if has_pris else ""))
    if self.tuning["so_mapping"] == "balanced":
        if self.tuning["so_loops"] == "dots":
            _so_emit_balanced_main_dots(self, tree=True, compact=False, local=local, blocks=blocks)
        else:
            _so_emit_balanced_main(self, tree=True)
        self.gen_add_end_function()
        return
    sub = lambda jv: "(c >= %s && c < %s + grid_so_tree_topology[%d*%s + 2])" % (jv, jv, K, jv)
    _so_emit(self, """
// main loops: joint m supplies the vectors this lane's operators act on; joint l, on the same root path, the vector the results are dotted with
#pragma unroll 1
for (int m = 0; m < @N@; m++) {""")
    _so_emit(self, _SO_OPERATORS)
    _so_emit(self, "    const int m_end = m + grid_so_tree_topology[%d*m + 2];\n    #pragma unroll 1\n    for (int l = m; l < m_end; l++) { // l in the subtree of m (incl. m): joint j = l, ancestor-or-self an = m" % K)
    _so_emit(self, _SO_X)
    _so_emit(self, "        {")
    _so_emit(self, _SO_CASE_A, SUB=sub("j"))
    _so_emit(self, "        }\n    }")
    _so_emit(self, "    #pragma unroll 1\n    for (int l = m; l >= 0; l = grid_so_tree_topology[%d*l]) { // l = m, then its ancestors: joint j = m, ancestor-or-self an = l" % K)
    _so_emit(self, _SO_X)
    _so_emit(self, "        {")
    _so_emit(self, _SO_CASE_B, SUBX="(c > j && c < j + grid_so_tree_topology[%d*j + 2])" % K)
    _so_emit(self, "        }\n        if (l != m) {")
    _so_emit(self, _SO_CASE_C, SUB=sub("j"))
    _so_emit(self, "        }\n    }\n}")
    self.gen_add_end_function()


def gen_idsva_so_device(self, use_thread_group=False, use_qdd_input=False, compact=False):
    n = self.model.n
    params = ["so is the compact staging record of this solve in LDS: IDSVA_SO_STAGE_PER_SOLVE values (see idsva_so_inner_compact)" if compact else
              "so is the output record of this solve: 4*NUM_JOINTS^3 values [d2tau_dq2 | d2tau_dqd2 | d2tau_dvdq | dM_dq] (global or LDS memory)",
              "s_q is the vector of joint positions in LDS", "s_qd is the vector of joint velocities in LDS"]
    if use_qdd_input:
        params.append("s_qdd is the vector of joint accelerations in LDS")
    params += ["s_scratch is LDS scratch of IDSVA_SO_SCRATCH_PER_SOLVE elements (X(q), then the per-joint records)", "d_robotModel is the pointer to the initialized model specific helpers on the GPU",
               "gravity is the gravity constant", "lane is the caller's lane index inside the solve's lane group", "active is false for lane groups without a solve"]
    self.gen_add_func_doc("Computes the second order derivatives of inverse dynamics: X(q) update + idsva_so_inner (lane-group cooperative)",
                          ["all lanes of the solve's lane group must call it" + ("" if use_qdd_input else "; qdd = 0")], params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__device__ __forceinline__")
    self.gen_add_code_line("void idsva_so_device" + ("_compact" if compact else "") + "(T *so, const T *s_q, const T *s_qd, " + ("const T *s_qdd, " if use_qdd_input else "") +
                           "T *s_scratch, const robotModel<T> *d_robotModel, const T gravity, const int lane, const bool active" +
                           (", const bool blocks_only = false" if (self.gen_idsva_so_mode() == "tree" and len(self.model.roots) > 1) else "") + ") {", True)
    self.gen_add_code_line("T *s_X = s_scratch;")
    if not use_qdd_input:
        self.gen_add_code_line("T *s_qdd = &s_scratch[%d];" % (self.gen_idsva_so_rec() * n))
        self.gen_add_code_line("if (lane < %d) { s_qdd[lane] = static_cast<T>(0); }" % n)
    self.gen_load_update_XImats_helpers_function_call(use_thread_group)
    self.gen_idsva_so_inner_function_call(use_thread_group, use_qdd_input, compact=compact)
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
                          ["launch with IDSVA_SO_SUGGESTED_THREADS threads and IDSVA_SO_DYNAMIC_SHARED_MEM_COUNT*sizeof(T) bytes of dynamic LDS" + (": the 4 n^3 record of every solve is staged in LDS" if not self.gen_idsva_so_direct() else "; the 4 n^3 record of a solve does not fit LDS and is written entry by entry to global memory")], func_params, None)
    self.gen_add_code_line("template <typename T>")
    self.gen_add_code_line("__global__ GRID_LAUNCH_BOUNDS")
    self.gen_add_code_line(func_def, True)
    sl, scratch, stage, threads = self.gen_idsva_so_lds_layout()
    pad3n = (3 * n + 3) // 4 * 4
    self.gen_kernel_prologue("IDSVA_SO_LDS_PER_SOLVE", "IDSVA_SO_MAX_SOLVES_PER_BLOCK")
    self.gen_add_code_line("T *s_q_qd_u = s_mem; T *s_q = s_q_qd_u; T *s_qd = &s_q_qd_u[%d]; T *s_qdd = &s_q_qd_u[%d]; T *s_scratch = &s_mem[%d]; (void)s_qdd;" % (n, 2 * n, pad3n))
    direct = self.gen_idsva_so_direct()
    compact = self.gen_idsva_so_packed()
    if not direct:
        self.gen_add_code_line("T *s_idsva_so = &s_out_all[grp*%d]; // this solve's output record, staged in LDS%s" % (stage, " in compact form (every value once)" if compact else ""))
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
    if direct:
        self.gen_add_code_line("// (the record of one solve is larger than the LDS of a CU: every entry goes straight to global memory)")
        self.gen_add_code_line("idsva_so_device<T>(&d_idsva_so[static_cast<size_t>(kc)*%d], s_q, s_qd, " % (4 * n3) + ("s_qdd, " if use_qdd_input else "") + "s_scratch, d_robotModel, gravity, lane, valid);")
        self.gen_add_sync(use_thread_group)
    else:
        self.gen_add_code_line("idsva_so_device" + ("_compact" if compact else "") + "<T>(s_idsva_so, s_q, s_qd, " + ("s_qdd, " if use_qdd_input else "") + "s_scratch, d_robotModel, gravity, lane, true);")
    if single_call_timing:
        self.gen_add_end_control_flow()
    if direct:
        pass
    elif single_call_timing:
        if compact:
            self.gen_add_sync(use_thread_group)
            self.gen_add_code_line("// save down to global: the dense record gathered from the compact staging")
            self.gen_add_parallel_loop("ind", str(4 * n3), use_thread_group)
            self.gen_add_code_line("d_idsva_so[ind] = s_idsva_so[grid_so_expand[ind]];")
            self.gen_add_end_control_flow()
        else:
            self.gen_kernel_save_result_single_timing("idsva_so", 4 * n3, use_thread_group)
    elif compact:
        if int(self.tuning["debug_stop"]) == 32:  # timing ablation (wrong results): everything computed and staged, the record never leaves the CU
            self.gen_add_code_line("if (gravity < static_cast<T>(-1e30)) {", True)
        self.gen_kernel_save_result_expanded("idsva_so", 4 * n3, stage, "grid_so_expand", use_thread_group)
        if int(self.tuning["debug_stop"]) == 32:
            self.gen_add_end_control_flow()
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
    if self.so_wide_lanes():
        self.gen_add_code_line("// (GRID_SO_WIDE) the kernels of the nested library for wider lane groups do the work; block_dimms may count lane groups of either width: the kernels grid-stride")
        self.gen_add_code_line("wide::" + name + "<T, USE_QDD_FLAG>(reinterpret_cast<wide::gridData<T> *>(hd_data), reinterpret_cast<const wide::robotModel<T> *>(d_robotModel), gravity, num_timesteps, block_dimms, thread_dimms" + ("" if compute_only else ", streams") + "); // (same layouts)")
        self.gen_add_end_function()
        return
    self.gen_add_code_line("int stride_q_qd = 3*NUM_JOINTS;")
    self.gen_add_code_line("if (num_timesteps > grid_so_max_timesteps<T>()) {gpuErrchk(hipErrorInvalidValue); return;} // (beyond what init_gridData allocates for second-order records)")
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
    if self.gen_idsva_so_packed():  # what the kernels run; the dense-record functions above stay for callers that hand in a record of their own
        self.gen_idsva_so_inner(use_thread_group, True, compact=True)
        self.gen_idsva_so_device(use_thread_group, False, compact=True)
        self.gen_idsva_so_device(use_thread_group, True, compact=True)
    for qdd in (True, False):
        for timing in (True, False):
            self.gen_idsva_so_kernel(use_thread_group, qdd, timing)
    for mode in (0, 1, 2):
        self.gen_idsva_so_host(mode)
