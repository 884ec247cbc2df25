"""Forward-dynamics gradient for branched robots with revolute joints: every BRANCH of the tree in the frame of its own tip link.

Generalises algorithms/_tip_frame_gradient.py (one frame for a serial chain) to kinematic trees; same function as the column walk
(reference algorithms/_inverse_dynamics_gradient.py:27-775, _direct_minv.py:23-453, _forward_dynamics_gradient.py:7-62; oracle
/root/reference/_test.py:229-520) and the same identities (Singh, Russell & Wensing, RA-L 2022; see the module notes of _tip_frame_gradient.py).

Why one frame per branch: the identities hold in ANY single inertial frame, but fp32 only survives the later M^-1 amplification if the
light distal links are described about a nearby origin (measured on the 7-DoF arm: 1.8e-4 in the world frame, 3e-6 in the tip frame).
A tree has several tips, so every maximal unbranched run of joints (a branch) uses the frame F_b of ITS last link.

  * lanes: the joints of a branch sit on consecutive lanes of one 16-lane DPP row (branches are bin-packed into the rows of the lane group,
    so lane != joint id in general; a per-lane table row gives joint id, position, branch length, tree level, ...)
  * frames: every lane walks the root path of its branch, tip -> root (wave-uniform per branch), with the rotation blocks of X(q):
    it keeps the pose of its own link, the pose T_b of the parent branch's frame (for the hand-over below) and the joint axes S_i of ALL
    joints on the path, expressed in F_b
  * kinematics need no cross-lane traffic at all: v, a of the own link and Pd_i = v_i x S_i, Pdd_i of every ancestor i are running sums along
    the path (root -> own joint), evaluated redundantly by the lanes in their own branch frame
  * composites I^C (10), B^C (12), f^C (6): suffix DPP scan inside the branch, then from the deepest tree level upwards the total of a
    branch is re-expressed in the parent branch's frame (T_b) and handed to ALL lanes of the parent branch through LDS
  * entries: lane k (joint k) evaluates, in F_b, everything that couples k with its ancestors-or-self i:
        M[i][k] = S_i . t1_k     dc_i/dq_k = S_i . t3_k     dc_i/dqd_k = S_i . t2_k      (column k, rows i)
        dc_k/dq_i = t1_k . Pdd_i + t4_k . Pd_i      dc_k/dqd_i = 2 t1_k . Pd_i + t4_k . S_i      (row k, columns i != k)
    into the (zero-filled) LDS image of dc/du and M; unrelated joints stay exactly zero
  * M is never inverted: per base-rooted component the tree-sparse U D U^T factorisation (leaves first: no fill-in, Featherstone's
    branch-induced sparsity) is evaluated wave-uniformly in registers, parked in LDS, and every lane solves for its two columns of dc/du.

Scope: fixed-base robots whose joints are all revolute, branches of at most 16 joints that can be packed into the lane group's DPP rows
(the tree-sparse M and its factors must fit the X(q) storage: 2 * sum(ancestors + 1) + 4 <= 20 n).  Everything else stays on the column walk.
The stand-alone kernels of such robots (inverse dynamics, its gradient, forward dynamics, M^-1) are subsets of the same emitter
(_emit_branch_inner(mode)); serial chains of 10 or more joints use the fused inner for forward_dynamics_gradient too (their factors
live in LDS here, the tip-frame inner replicates a dense factorisation in registers).
"""
import numpy as np

from ._tip_frame_gradient import _probe


def gen_branch_frame_plan(self):
    """Decompose the tree into branches, pack them into DPP rows, derive the per-lane tables.  Returns None when the robot is out of scope."""
    m = self.model
    n = m.n
    lanes = self.lanes_per_solve
    if any(s_ >= 3 for s_ in m.S_index):
        return None
    parent = list(m.parent)
    ch = [list(m.children[j]) for j in range(n)]
    br = [-1] * n
    branches = []
    rowcap = min(16, lanes)
    for j in range(n):
        p = parent[j]
        if p < 0 or len(ch[p]) >= 2 or len(branches[br[p]]) >= rowcap:  # (a run longer than a DPP row continues as a child branch: a junction with one child)
            br[j] = len(branches)
            branches.append([j])
        else:
            if p != j - 1:
                return None  # ids are DFS pre-order: an only child follows its parent immediately
            br[j] = br[p]
            branches[br[j]].append(j)
    nb = len(branches)
    pb = [(-1 if parent[J[0]] < 0 else br[parent[J[0]]]) for J in branches]
    level = [0] * nb
    for b in range(nb):
        if pb[b] >= 0:
            level[b] = level[pb[b]] + 1
    paths = []
    for J in branches:
        path, j = [], J[-1]
        while j >= 0:
            path.append(j)
            j = parent[j]
        paths.append(path)
    rows = [[] for _ in range(lanes // rowcap)]
    for b in sorted(range(nb), key=lambda b_: (-len(branches[b_]), b_)):  # first-fit decreasing
        for r in rows:
            if sum(len(branches[x]) for x in r) + len(branches[b]) <= rowcap:
                r.append(b)
                break
        else:
            return None
    joint_of_lane = [-1] * lanes
    for ri, r in enumerate(rows):
        at = ri * rowcap
        for b in sorted(r):
            for j in branches[b]:
                joint_of_lane[at] = j
                at += 1
    kids = [[c for c in range(nb) if pb[c] == b] for b in range(nb)]
    # base-rooted components (contiguous id ranges) and their shapes
    comp_base = [0] * n
    for j in range(n):
        comp_base[j] = j if parent[j] < 0 else comp_base[parent[j]]
    comps = sorted(set(comp_base))
    shapes, shape_of = [], {}
    for cb in comps:
        size = len(m.subtree[cb])
        sig = tuple((parent[cb + i] - cb if parent[cb + i] >= 0 else -1) for i in range(size))
        if sig not in shapes:
            shapes.append(sig)
        shape_of[cb] = shapes.index(sig)
    D = max(len(p_) for p_ in paths)
    maxchild = max([len(k_) for k_ in kids] + [0])
    ncross = len(set(i for p_ in paths for i in range(1, len(p_)) if br[p_[i]] != br[p_[i - 1]]))  # path steps at which some root path enters a parent branch
    owner = branch_owner_walk(self, D, max(level), ncross)
    row_len = (24 + D + max(maxchild, 1) + 3 * D + (D if owner else 0) + 3) // 4 * 4 + 4  # (header | path codes | child branches | joint offsets along the path | [lanes of the path joints] | parent branch slot, lane of the parent link, pad)
    # compact (tree-sparse) storage of M and of its factors: column k holds the entries of its ancestors (ascending) and then the diagonal
    # (every base-rooted component starts at a multiple of 4 values: the column solves read a component's factors with 16-byte LDS loads)
    mstart, at = [0] * n, 0
    for j in range(n):
        if parent[j] < 0:
            at = (at + 3) // 4 * 4
        mstart[j] = at
        at += len(m.ancestors[j]) + 1
    at = (at + 3) // 4 * 4
    ubase = {cb: mstart[cb] for cb in comps}
    # multiply-adds of the (replicated) factorisation of the largest component: the price of keeping it wave-uniform
    factor_work = max(sum(len(m.ancestors[cb + i]) * (len(m.ancestors[cb + i]) + 1) // 2 for i in range(len(m.subtree[cb]))) for cb in comps)
    nnz = at
    # LDS placement: once the frames are known the X(q) storage (20 n values) is re-used for M (must fit, with one spare quad for the
    # branch-free stores), then for the factors and the branch hand-over records as far as they fit; what does not goes behind the path axes
    if nnz + 4 > 20 * n:
        return None
    place, x_at, sp_at = {"M": ("x", 0), "trash": ("x", nnz)}, nnz + 4, (0 if owner else 6 * D * nb)  # (owner walk: no path records at all)
    for item, size in (("U", nnz), ("G", 28 * nb)):
        if x_at + size <= 20 * n:
            place[item] = ("x", x_at)
            x_at += size
        else:
            sp_at = (sp_at + 3) // 4 * 4
            place[item] = ("sp", sp_at)
            sp_at += size
    # one spare set of path records at the very end takes the (branch-free) stores of the lanes that are not the first of their branch; the
    # kernels that have a result image in LDS send those stores to the head of the image instead (it is cleared after the frame chain), so
    # the forward-dynamics-gradient kernel does not carry the spare set in its LDS slice
    sp_spare, sp_at = sp_at, sp_at + (0 if owner else 6 * D)
    return dict(owner=owner, sp_spare=sp_spare, branches=branches, br=br, pb=pb, level=level, paths=paths, joint_of_lane=joint_of_lane, kids=kids, comp_base=comp_base,
                shapes=shapes, shape_of=shape_of, D=D, maxLb=max(len(J) for J in branches), maxlevel=max(level), maxchild=maxchild, row_len=row_len, nb=nb, ubase=ubase, mstart=mstart, nnz=nnz, place=place, sp_size=sp_at, factor_work=factor_work)


def branch_spare_in_image(self, image_len):
    """The lanes that are not the first of their branch send their (branch-free) path-record stores of the frame chain to the head of the
    kernel's result image when that is large enough and one quad per lane clears it again; else to a spare set of records in s_SP."""
    D = self.branch_plan["D"]
    if self.branch_plan["owner"]:
        return False  # (no path records, no spare set)
    return image_len >= 6 * D and (6 * D + 3) // 4 <= self.lanes_per_solve


def branch_owner_walk(self, D=None, maxlevel=None, ncross=0):
    """tuning branch_walk: path | owner | auto.  path = every lane walks the root path of its branch and re-derives the ancestors' S_i, Pd_i, Pdd_i in ITS
    branch frame by running sums (no cross-lane traffic, D steps of ~70 instructions, the joint axes of every path parked in LDS).  owner = every lane
    keeps S, Pd, Pdd of its own joint in its own branch frame (velocities and accelerations by prefix scans inside the branch plus one hand-down per tree
    level), the entries that couple joint k with an ancestor i are dot products in the frame of i's branch: lane k fetches the owner's vectors with
    cross-lane reads (ds_bpermute) and re-expresses ITS t-vectors once per branch crossing; frames by the transform scan; no path records in LDS."""
    want = self.tuning["branch_walk"]
    if want not in ("path", "owner", "auto"):
        raise ValueError("tuning['branch_walk'] must be path, owner or auto")
    if D is None:
        return self.branch_plan["owner"]
    if want != "auto":
        return want == "owner"
    # instructions a wave spends on the ancestors, per path step and phase (static counts of the 30-DoF humanoid, tools/isa_stats.py): the path walk
    # pays ~110 per step in its walks and ~80 in the frame chain (which the long single chains already replaced by the scan); the owner walk ~48
    # per step plus the scan, the hand-downs per tree level and the re-expression of the t-vectors per crossing step.  Measured per 16 384 solves
    # (profiles/ab/r3_owner_walk.jsonl): humanoid (D 10, two levels) 87.1 -> 77.5 us, 12-joint chain (D 12) 39.1 -> 34.2, 12-DoF tree (D 6, three levels) 30.8 -> 34.2
    scan_default = maxlevel == 0 and D >= 10
    path_cost = D * (110 + (0 if scan_default else 80))
    owner_cost = 280 + 110 * maxlevel + 105 * ncross + 48 * D
    return 1.15 * owner_cost < path_cost


def branch_factor_by_branch(self):
    """tuning factor_split: "branch" = every branch eliminates its own pivots (lanes of different branches work in parallel, tree level by tree
    level, Schur complements onto the ancestors handed up through the 28-value branch records); "component" = every lane factors its whole
    base-rooted component.  auto = branch where the hand-over fits the records (a0 (a0 + 3) / 2 <= 28 values, a0 = ancestors above the branch)
    and the split removes at least a quarter of the multiply-adds a wave executes."""
    want = self.tuning["factor_split"]
    P, m = self.branch_plan, self.model
    a0 = lambda b: len(P["paths"][b]) - len(P["branches"][b])
    fits = all(a0(b) * (a0(b) + 3) // 2 <= 28 for b in range(P["nb"]))
    if want == "branch" and not fits:
        raise NotImplementedError("factor_split=branch: a branch of this robot has more than 6 ancestors above it (the hand-over record holds 28 values)")
    if want != "auto":
        return want == "branch"
    # multiply-adds one wave executes (all code paths of a wave run, exec-masked): one block per component shape against one block per
    # (level, branch length, path length).  Measured, 16 384 solves: 30-DoF humanoid 367 -> 206, 109.6 -> 97.3 us (and no scratch any more);
    # 12-DoF tree 65 -> 64 with two more hand-over levels, 31.1 -> 33.1 us
    tri = lambda k: k * (k + 1) // 2
    shape_work = {}
    for j in range(m.n):
        sh = P["shape_of"][P["comp_base"][j]]
        shape_work.setdefault(sh, {}).setdefault(P["comp_base"][j], 0)
        shape_work[sh][P["comp_base"][j]] += tri(len(m.ancestors[j]))
    by_component = sum(max(w.values()) for w in shape_work.values())
    by_branch = sum({(P["level"][b], len(P["branches"][b]), len(P["paths"][b])): sum(tri(k) for k in range(a0(b), len(P["paths"][b]))) for b in range(P["nb"])}.values())
    return fits and 4 * by_branch <= 3 * by_component


def _emit_factor_by_branch(self, P, with_rhs, R32, use_thread_group, ptr):
    """Tree-sparse U D U^T, distributed over the branches (see branch_factor_by_branch).  Path positions d = 0 .. plen-1 count from the component's
    root; a branch of Lb joints below a0 = plen - Lb ancestors owns the columns d >= a0 of the triangle W[i][j], i <= j < plen."""
    A = self.gen_add_code_line
    nb, branches, paths, level, kids = P["nb"], P["branches"], P["paths"], P["level"], P["kids"]
    maxlevel, maxchild, maxLb = P["maxlevel"], P["maxchild"], P["maxLb"]
    tri = lambda i, j: j * (j + 1) // 2 + i  # packed index of (i <= j) in a hand-over record
    A("// tree-sparse U D U^T of the joint-space inertia, leaves first (no fill-in), distributed over the branches: the lanes of a branch eliminate the")
    A("// pivots of their own joints (replicated inside the branch only), tree level by tree level; the Schur complement onto the ancestors and the")
    A("// forward-substituted right-hand side travel to the parent branch through the branch records; the first lane of a branch parks its factors")
    A("T *s_Ub = %s, *s_Mb = %s; // (absolute: column of joint j starts at mstart(j))" % (ptr("U"), ptr("M")))
    A("const int a0 = plen - Lb, mb0 = mstart - pos*a0 - (pos*(pos + 1))/2, jid0 = jid - pos; // ancestors above the branch, column start and id of the branch's first joint")
    A("(void)a0; (void)mb0; (void)jid0; (void)ubase; (void)li; (void)cbase; (void)shape;")
    if with_rhs:
        A("T bk[%d]; // forward-substituted right-hand side of this branch's joints" % maxLb)
        A("#pragma unroll")
        A("for (int r = 0; r < %d; r++) { bk[r] = Z; }" % maxLb)
    sigs = {}
    for b in range(nb):
        sigs.setdefault(level[b], {}).setdefault((len(branches[b]), len(paths[b])), []).append(b)
    col0 = lambda a0_, t: sum(a0_ + s_ + 1 for s_ in range(t))  # offset of the column of the branch's joint t behind mb0
    for lv in range(maxlevel, -1, -1):
        first = True
        for (Lb_, pl), bs in sorted(sigs.get(lv, {}).items()):
            a0_ = pl - Lb_
            A("%sif (level == %d && Lb == %d && plen == %d) { // branches of %d joints below %d ancestors" % ("" if first else "else ", lv, Lb_, pl, Lb_, a0_), True)
            first = False
            for t in range(Lb_):
                k = a0_ + t
                A(" ".join("T W%d_%d = s_Mb[mb0 + %d];" % (i, k, col0(a0_, t) + i) for i in range(k + 1)))
            if a0_:
                A("T " + ", ".join("W%d_%d = Z" % (i, j) for j in range(a0_) for i in range(j + 1)) + "; // Schur complement onto the ancestors")
            if with_rhs:
                if a0_:
                    A("T " + ", ".join("b%d = Z" % d for d in range(a0_)) + ";")
                A(" ".join("T b%d = s_qdd[jid0 + %d];" % (a0_ + t, t) for t in range(Lb_)))
            if any(kids[b] for b in bs):
                for c in range(maxchild):
                    A("if (cs%d >= 0) { const T *g = &s_G[28*cs%d]; // what child branch %d leaves for its ancestors" % (c, c, c), True)
                    A(" ".join("W%d_%d += g[%d];" % (i, j, tri(i, j)) for j in range(pl) for i in range(j + 1)))
                    if with_rhs:
                        A(" ".join("b%d += g[%d];" % (d, pl * (pl + 1) // 2 + d) for d in range(pl)))
                    self.gen_add_end_control_flow()
            for k in range(pl - 1, a0_ - 1, -1):
                A("const T rd%d = %s;" % (k, R32("grid_rcp(W%d_%d)" % (k, k))))
                if k:
                    A(" ".join("const T U%d_%d = %s;" % (i, k, R32("W%d_%d*rd%d" % (i, k, k))) for i in range(k)))
                    for j in range(k):
                        A(" ".join("W%d_%d -= U%d_%d*W%d_%d;" % (i, j, i, k, j, k) for i in range(j + 1)))
                    if with_rhs:
                        A(" ".join("b%d -= U%d_%d*b%d;" % (i, i, k, k) for i in range(k)))
                if with_rhs:
                    A("b%d *= rd%d;" % (k, k))
            if with_rhs:
                A(" ".join("bk[%d] = b%d;" % (t, a0_ + t) for t in range(Lb_)))
            A("if (pos == 0) { // the first lane of the branch parks the factors%s" % (" and posts the hand-over" if a0_ else ""), True)
            for t in range(Lb_):
                k = a0_ + t
                A(" ".join(["s_Ub[mb0 + %d] = U%d_%d;" % (col0(a0_, t) + i, i, k) for i in range(k)] + ["s_Ub[mb0 + %d] = rd%d;" % (col0(a0_, t) + k, k)]))
            if a0_:
                A("T *g = &s_G[28*slot];")
                A(" ".join("g[%d] = W%d_%d;" % (tri(i, j), i, j) for j in range(a0_) for i in range(j + 1)))
                if with_rhs:
                    A(" ".join("g[%d] = b%d;" % (a0_ * (a0_ + 1) // 2 + d, d) for d in range(a0_)))
            self.gen_add_end_control_flow()
            self.gen_add_end_control_flow()
        self.gen_add_sync(use_thread_group)
    if not with_rhs:
        return
    A("// back substitution, root branches first: qdd of the ancestors comes back through s_qdd, the factors from where they were parked")
    for lv in range(0, maxlevel + 1):
        first = True
        for (Lb_, pl), bs in sorted(sigs.get(lv, {}).items()):
            a0_ = pl - Lb_
            A("%sif (level == %d && Lb == %d && plen == %d) {" % ("" if first else "else ", lv, Lb_, pl), True)
            first = False
            if a0_:
                A(" ".join("const T x%d = s_qdd[pj%d];" % (d, pl - 1 - d) for d in range(a0_)))
            for t in range(Lb_):
                k = a0_ + t
                A("const T x%d = bk[%d]%s;" % (k, t, "".join(" - s_Ub[mb0 + %d]*x%d" % (col0(a0_, t) + i, i) for i in range(k))))
            A("if (pos == 0) { " + " ".join("s_qdd[jid0 + %d] = x%d;" % (t, a0_ + t) for t in range(Lb_)) + " }")
            self.gen_add_end_control_flow()
        self.gen_add_sync(use_thread_group)


def branch_chain_scan(self):
    """tuning branch_chain: walk | scan | auto.  Measured per 16 384 solves: 12-joint chain (one branch, D = 12) walk 42.4 us, scan 38.7; 30-DoF humanoid
    (two tree levels, D = 10) 86.4 / 91.7; 12-DoF tree (three levels, D = 6) 30.7 / 32.5 - the scan pays for long single branches only (the level phases
    and the cross-lane broadcasts of the tree cost more than the shorter walks save)."""
    want = self.tuning["branch_chain"]
    if want not in ("walk", "scan", "auto"):
        raise ValueError("tuning['branch_chain'] must be walk, scan or auto")
    P = self.branch_plan
    return want == "scan" or (want == "auto" and P["maxlevel"] == 0 and P["D"] >= 10)


def _emit_chain_scan(self, P, H, RL, bfam, kin, use_thread_group, ptr, owner=False):
    """Frames of the branch-frame path without the per-lane walk (tuning branch_chain = scan): a log-step DPP scan of rigid transforms over the lanes of a
    branch gives every lane the pose of its own link in the branch frame; the pose of the parent branch's frame follows from the branch's first lane;
    the joint axes along the root path are written by the lanes that own them and, tree level by tree level, re-expressed by the child branches (one
    ancestor per lane) - instead of every lane stepping through all D path joints."""
    A = self.gen_add_code_line
    D, maxLb, maxlevel, maxchild = P["D"], P["maxLb"], P["maxlevel"], P["maxchild"]
    ro = H + D + max(maxchild, 1)
    one = "static_cast<T>(1)"
    A("{", True)
    A("const bool has_next = active && (pos + 1 < Lb); // a joint of the same branch follows towards the tip (ids are consecutive inside a branch)")
    A("const int jn = has_next ? jid + 1 : js, io = active ? own : 0, inx = has_next ? own - 1 : 0;")
    A("T Eo[9], En[9], ro_[3], rn_[3]; // E(q) and frame origin (in the parent's coordinates) of this lane's joint and of the next one")
    A("#pragma unroll")
    A("for (int r = 0; r < 9; r++) { Eo[r] = s_X[GRID_X_STRIDE*js + r]; En[r] = s_X[GRID_X_STRIDE*jn + r]; }")
    A("#pragma unroll")
    A("for (int r = 0; r < 3; r++) { ro_[r] = d_L[%d + 3*io + r]; rn_[r] = d_L[%d + 3*inx + r]; }" % (ro, ro))
    if not owner:
        A("const int pslot = static_cast<int>(d_L[%d]); // slot of the parent branch" % (RL - 4))
    self.gen_add_sync(use_thread_group)
    A("// (every lane has read X(q): its storage is free from here on)  pose of this link relative to the next one: (E_next, -E_next r_next); identity on tip lanes")
    A("T R[9], p[3];")
    A("#pragma unroll")
    A("for (int r = 0; r < 9; r++) { R[r] = has_next ? En[r] : ((r %% 4 == 0) ? %s : Z); }" % one)
    A("#pragma unroll")
    A("for (int r = 0; r < 3; r++) { p[r] = has_next ? -(En[3*r]*rn_[0] + En[3*r + 1]*rn_[1] + En[3*r + 2]*rn_[2]) : Z; }")
    for k in [k_ for k_ in (1, 2, 4, 8) if k_ < maxLb]:
        A("{ // suffix scan over the lanes of the branch, step %d: (R, p) <- (R, p)[lane + %d] o (R, p)" % (k, k), True)
        A("T Ra[9], pa[3], Rn[9], pn[3];")
        A("#pragma unroll")
        A("for (int r = 0; r < 9; r++) { Ra[r] = grid_lane_above<%d>(R[r]); }" % (k * (2 if getattr(self, "lane_interleave", False) else 1)))
        A("#pragma unroll")
        A("for (int r = 0; r < 3; r++) { pa[r] = grid_lane_above<%d>(p[r]); }" % (k * (2 if getattr(self, "lane_interleave", False) else 1)))
        A("const bool ok = active && (pos + %d < Lb);" % k)
        A("#pragma unroll")
        A("for (int r = 0; r < 3; r++) {", True)
        A("#pragma unroll")
        A("for (int c = 0; c < 3; c++) { Rn[3*r + c] = Ra[3*r]*R[c] + Ra[3*r + 1]*R[3 + c] + Ra[3*r + 2]*R[6 + c]; }")
        A("pn[r] = pa[r] + Ra[3*r]*p[0] + Ra[3*r + 1]*p[1] + Ra[3*r + 2]*p[2];")
        self.gen_add_end_control_flow()
        A("#pragma unroll")
        A("for (int r = 0; r < 9; r++) { R[r] = ok ? Rn[r] : R[r]; }")
        A("#pragma unroll")
        A("for (int r = 0; r < 3; r++) { p[r] = ok ? pn[r] : p[r]; }")
        self.gen_add_end_control_flow()
    A("#pragma unroll")
    A("for (int r = 0; r < 9; r++) { myR[r] = R[r]; }")
    A("#pragma unroll")
    A("for (int r = 0; r < 3; r++) { myp[r] = p[r]; }")
    A("// pose of the parent branch's frame (the frame of the link this branch hangs off) in the branch frame: the first lane's pose composed with its own joint")
    A("const int l0 = lane - pos; // first lane of this lane's branch")
    A("{", True)
    A("T cR[9], cp[3];")
    A("#pragma unroll")
    A("for (int r = 0; r < 3; r++) {", True)
    A("#pragma unroll")
    A("for (int c = 0; c < 3; c++) { cR[3*r + c] = R[3*r]*Eo[c] + R[3*r + 1]*Eo[3 + c] + R[3*r + 2]*Eo[6 + c]; }")
    self.gen_add_end_control_flow()
    A("#pragma unroll")
    A("for (int r = 0; r < 3; r++) { cp[r] = p[r] - (cR[3*r]*ro_[0] + cR[3*r + 1]*ro_[1] + cR[3*r + 2]*ro_[2]); }")
    A("#pragma unroll")
    A("for (int r = 0; r < 9; r++) { TR[r] = grid_group_shfl(cR[r], l0); }")
    A("#pragma unroll")
    A("for (int r = 0; r < 3; r++) { Tp[r] = grid_group_shfl(cp[r], l0); }")
    self.gen_add_end_control_flow()
    if bfam:
        A("{ const int lB = (active && iB >= 0) ? l0 + Lb - 1 - iB : lane; // the lane of the branch's middle joint")
        A("  pB[0] = grid_group_shfl(myp[0], lB); pB[1] = grid_group_shfl(myp[1], lB); pB[2] = grid_group_shfl(myp[2], lB);")
        A("  if (!(active && iB >= 0)) { pB[0] = pB[1] = pB[2] = Z; } }")
    if owner:  # no path records: the entries are evaluated in the frames of the ancestors' branches (branch_owner_walk)
        if kin:
            A("if (level == 0) { gvec[0] = gravity*TR[2]; gvec[1] = gravity*TR[5]; gvec[2] = gravity*TR[8]; } // base acceleration (0,0,g) in the coordinates of a branch that hangs off the base (deeper branches receive it with the parent link's acceleration)")
        else:
            A("(void)gvec;")
        self.gen_add_end_control_flow()
        return
    A("T *s_Gh = %s; // branch records: free until the composites are handed over (here: gravity direction per branch)" % ptr("G"))
    A("(void)s_Gh; (void)pslot;")
    for lv in range(maxlevel + 1):
        A("if (active && level == %d) { // tree level %d: joint axes along the root path, in this branch's frame" % (lv, lv), True)
        if kin:
            if lv == 0:
                A("gvec[0] = gravity*TR[2]; gvec[1] = gravity*TR[5]; gvec[2] = gravity*TR[8]; // base acceleration (0,0,g) in this branch's coordinates")
            else:
                A("{ const T *gp = &s_Gh[28*pslot];")
                A("  #pragma unroll")
                A("  for (int r = 0; r < 3; r++) { gvec[r] = TR[3*r]*gp[0] + TR[3*r + 1]*gp[1] + TR[3*r + 2]*gp[2]; } }")
            if lv < maxlevel:
                A("if (pos == 0) { s_Gh[28*slot] = gvec[0]; s_Gh[28*slot + 1] = gvec[1]; s_Gh[28*slot + 2] = gvec[2]; }")
        A("{ // this lane's own joint")
        A("  const int ax = static_cast<int>(Lc[11]); T w[3];")
        A("  #pragma unroll")
        A("  for (int r = 0; r < 3; r++) { w[r] = (ax == 0) ? myR[3*r] : ((ax == 1) ? myR[3*r + 1] : myR[3*r + 2]); }")
        A("  T *rec = &s_Sp[6*own];")
        A("  rec[0] = w[0]; rec[1] = w[1]; rec[2] = w[2];")
        A("  rec[3] = myp[1]*w[2] - myp[2]*w[1]; rec[4] = myp[2]*w[0] - myp[0]*w[2]; rec[5] = myp[0]*w[1] - myp[1]*w[0]; }")
        if lv > 0:
            A("#pragma unroll 1")
            A("for (int k = pos; k < plen - Lb; k += Lb) { // the ancestors beyond the branch: the parent branch's records, re-expressed (one per lane and trip)", True)
            A("const T *src = &s_SP[%d*pslot + 6*k]; T *dst = &s_Sp[6*(Lb + k)];" % (6 * D))
            A("T w[3], mo[3];")
            A("#pragma unroll")
            A("for (int r = 0; r < 3; r++) { w[r] = TR[3*r]*src[0] + TR[3*r + 1]*src[1] + TR[3*r + 2]*src[2]; mo[r] = TR[3*r]*src[3] + TR[3*r + 1]*src[4] + TR[3*r + 2]*src[5]; }")
            A("dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2];")
            A("dst[3] = mo[0] + Tp[1]*w[2] - Tp[2]*w[1]; dst[4] = mo[1] + Tp[2]*w[0] - Tp[0]*w[2]; dst[5] = mo[2] + Tp[0]*w[1] - Tp[1]*w[0];")
            self.gen_add_end_control_flow()
        A("#pragma unroll 1")
        A("for (int k = plen + pos; k < %d; k += Lb) { // zero beyond the root" % D, True)
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { s_Sp[6*k + r] = Z; }")
        self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
        self.gen_add_sync(use_thread_group)
    if not kin:
        A("(void)gvec;")
    self.gen_add_end_control_flow()


def gen_branch_frame_constants(self):
    """Table rows appended to grid_model_constants: one row per lane
    [(last quad of the row: slot of the parent branch) Ic (6) | c (3) | m | damping | axis | joint id | pos | branch length | level | branch slot | component base | shape | path length | factor base of the component | start of the joint's column in the compact M | path step of the branch's base-origin joint (-1: none) | 1 if this lane's column of M uses the base family |
     path codes (4*joint + axis, tip -> root, -1 = none) x D | child branch slots (-1 = none) x maxchild |
     origin of every path joint's frame in its parent's coordinates x D (so that no table read depends on another one)]."""
    m = self.model
    P = self.branch_plan
    rows = []
    for lane in range(self.lanes_per_solve):
        j = P["joint_of_lane"][lane]
        row = [0.0] * P["row_len"]
        if j < 0:
            row[12] = -1.0
            row[18] = -1.0
            for i in range(P["D"]):
                row[24 + i] = -1.0
            for c in range(max(P["maxchild"], 1)):
                row[24 + P["D"] + c] = -1.0
            if P["owner"]:
                for i in range(P["D"]):
                    row[24 + P["D"] + max(P["maxchild"], 1) + 3 * P["D"] + i] = -1.0
                row[P["row_len"] - 3] = -1.0
            rows += row
            continue
        I = m.I[j]
        mass = I[3, 3]
        H = I[:3, 3:]
        h = np.array([H[2, 1], H[0, 2], H[1, 0]])
        c = h / mass
        Ic = I[:3, :3] - mass * (float(c @ c) * np.eye(3) - np.outer(c, c))
        b = P["br"][j]
        J = P["branches"][b]
        row[:12] = [Ic[0, 0], Ic[0, 1], Ic[0, 2], Ic[1, 1], Ic[1, 2], Ic[2, 2], c[0], c[1], c[2], mass, float(m.damping[j]), float(m.S_index[j])]
        row[12:20] = [float(j), float(J.index(j)), float(len(J)), float(P["level"][b]), float(b), float(P["comp_base"][j]),
                      float(P["shape_of"][P["comp_base"][j]]), float(len(P["paths"][b]))]
        row[20] = float(P["ubase"][P["comp_base"][j]])
        row[21] = float(P["mstart"][j])
        # base-origin family (see _tip_frame_gradient._emit_base_family): leaf branches of 5 or more joints evaluate the columns of M of their
        # base half about the frame origin of the branch's middle joint instead of the branch tip
        jB = len(J) // 2
        use_b = self.tuning["base_origin"] != "off" and len(J) >= 5 and not P["kids"][b]
        row[22] = float(len(J) - 1 - jB) if use_b else -1.0
        row[23] = 1.0 if (use_b and J.index(j) <= jB) else 0.0
        for i in range(P["D"]):
            path = P["paths"][b]
            row[24 + i] = float(4 * path[i] + m.S_index[path[i]]) if i < len(path) else -1.0
        for c_ in range(max(P["maxchild"], 1)):
            row[24 + P["D"] + c_] = float(P["kids"][b][c_]) if c_ < len(P["kids"][b]) else -1.0
        ro = 24 + P["D"] + max(P["maxchild"], 1)
        for i, pj in enumerate(P["paths"][b]):
            r = self.gen_tip_frame_joint_offset(pj)
            row[ro + 3 * i:ro + 3 * i + 3] = [float(r[0]), float(r[1]), float(r[2])]
        if P["owner"]:  # lane that owns every path joint (+64: the path enters the parent branch here), lane of the link the branch hangs off
            lane_of = {jj: ll for ll, jj in enumerate(P["joint_of_lane"]) if jj >= 0}
            path = P["paths"][b]
            for i in range(P["D"]):
                row[ro + 3 * P["D"] + i] = float(lane_of[path[i]] + (64 if (i > 0 and P["br"][path[i]] != P["br"][path[i - 1]]) else 0)) if i < len(path) else -1.0
            row[P["row_len"] - 3] = float(lane_of[path[len(J)]]) if len(path) > len(J) else -1.0
        row[P["row_len"] - 4] = float(P["pb"][b])  # slot of the parent branch (-1: the branch hangs off the base)
        rows += row
    return rows


_BRANCH_LIBRARY = r"""
// ---------------------------------------------------------------------------------------------------------------------
// branch-frame gradient path (trees of revolute joints): hand-over of composite quantities from a branch frame to its parent's
// ---------------------------------------------------------------------------------------------------------------------
// y = R^T x  (R maps parent-frame coordinates to branch-frame coordinates)
template <typename T>
__device__ __forceinline__ void grid_rt3(T *y, const T (&R)[9], const T x0, const T x1, const T x2) {
    y[0] = R[0]*x0 + R[3]*x1 + R[6]*x2;
    y[1] = R[1]*x0 + R[4]*x1 + R[7]*x2;
    y[2] = R[2]*x0 + R[5]*x1 + R[8]*x2;
}
// y = R^T A R for symmetric A = [xx xy xz yy yz zz]
template <typename T>
__device__ __forceinline__ void grid_rt_sym(T *y, const T (&R)[9], const T (&A)[6]) {
    T W[9];
    #pragma unroll
    for (int c = 0; c < 3; c++) {
        W[c]     = A[0]*R[c] + A[1]*R[3 + c] + A[2]*R[6 + c];
        W[3 + c] = A[1]*R[c] + A[3]*R[3 + c] + A[4]*R[6 + c];
        W[6 + c] = A[2]*R[c] + A[4]*R[3 + c] + A[5]*R[6 + c];
    }
    y[0] = R[0]*W[0] + R[3]*W[3] + R[6]*W[6];
    y[1] = R[0]*W[1] + R[3]*W[4] + R[6]*W[7];
    y[2] = R[0]*W[2] + R[3]*W[5] + R[6]*W[8];
    y[3] = R[1]*W[1] + R[4]*W[4] + R[7]*W[7];
    y[4] = R[1]*W[2] + R[4]*W[5] + R[7]*W[8];
    y[5] = R[2]*W[2] + R[5]*W[5] + R[8]*W[8];
}
// force vector [n; f] about the origin of the branch frame -> about the origin p of the parent's frame, in the parent's coordinates
template <typename T>
__device__ __forceinline__ void grid_force_up(T *y, const T (&R)[9], const T (&p)[3], const T (&f)[6]) {
    grid_rt3(y, R, f[0] - (p[1]*f[5] - p[2]*f[4]), f[1] - (p[2]*f[3] - p[0]*f[5]), f[2] - (p[0]*f[4] - p[1]*f[3]));
    grid_rt3(y + 3, R, f[3], f[4], f[5]);
}
// motion vector [w; v_O] given in the parent's frame (about its origin) -> in the branch frame (about its origin): w' = R w, v' = R v_O + p x w'
template <typename T>
__device__ __forceinline__ void grid_motion_down(T (&y)[6], const T (&R)[9], const T (&p)[3], const T (&x)[6]) {
    #pragma unroll
    for (int r = 0; r < 3; r++) { y[r] = R[3*r]*x[0] + R[3*r + 1]*x[1] + R[3*r + 2]*x[2]; y[3 + r] = R[3*r]*x[3] + R[3*r + 1]*x[4] + R[3*r + 2]*x[5]; }
    y[3] += p[1]*y[2] - p[2]*y[1]; y[4] += p[2]*y[0] - p[0]*y[2]; y[5] += p[0]*y[1] - p[1]*y[0];
}
// composite inertia (10) of a whole branch, re-expressed in the parent branch's frame: origin shift (c' = c - p), then rotation
template <typename T>
__device__ __forceinline__ void grid_inertia_up(T *y, const T (&R)[9], const T (&p)[3], const T (&I)[10]) {
    {
        const T m = I[9];
        const T s = static_cast<T>(-2)*(I[6]*p[0] + I[7]*p[1] + I[8]*p[2]) + m*(p[0]*p[0] + p[1]*p[1] + p[2]*p[2]);
        T A[6];
        A[0] = I[0] + s + static_cast<T>(2)*I[6]*p[0] - m*p[0]*p[0];
        A[1] = I[1] + I[6]*p[1] + p[0]*I[7] - m*p[0]*p[1];
        A[2] = I[2] + I[6]*p[2] + p[0]*I[8] - m*p[0]*p[2];
        A[3] = I[3] + s + static_cast<T>(2)*I[7]*p[1] - m*p[1]*p[1];
        A[4] = I[4] + I[7]*p[2] + p[1]*I[8] - m*p[1]*p[2];
        A[5] = I[5] + s + static_cast<T>(2)*I[8]*p[2] - m*p[2]*p[2];
        grid_rt_sym(&y[0], R, A);
        grid_rt3(&y[6], R, I[6] - m*p[0], I[7] - m*p[1], I[8] - m*p[2]);
        y[9] = m;
    }
}
// composite Coriolis matrix [Sym | n | l] (12): Sym' = Sym + l p^T + p l^T - 2 (p.l) 1,  n' = n - p x l,  l' = l, then rotation
template <typename T>
__device__ __forceinline__ void grid_coriolis_up(T *y, const T (&R)[9], const T (&p)[3], const T (&B)[12]) {
    {
        const T l0 = B[9], l1 = B[10], l2 = B[11];
        const T s = static_cast<T>(-2)*(p[0]*l0 + p[1]*l1 + p[2]*l2);
        T A[6];
        A[0] = B[0] + s + static_cast<T>(2)*l0*p[0];
        A[1] = B[1] + l0*p[1] + p[0]*l1;
        A[2] = B[2] + l0*p[2] + p[0]*l2;
        A[3] = B[3] + s + static_cast<T>(2)*l1*p[1];
        A[4] = B[4] + l1*p[2] + p[1]*l2;
        A[5] = B[5] + s + static_cast<T>(2)*l2*p[2];
        grid_rt_sym(&y[0], R, A);
        grid_rt3(&y[6], R, B[6] - (p[1]*l2 - p[2]*l1), B[7] - (p[2]*l0 - p[0]*l2), B[8] - (p[0]*l1 - p[1]*l0));
        grid_rt3(&y[9], R, l0, l1, l2);
    }
}
// all three: y = [I^C (10) | B^C (12) | f^C (6)]
template <typename T>
__device__ __forceinline__ void grid_junction_up(T (&y)[28], const T (&R)[9], const T (&p)[3], const T (&I)[10], const T (&B)[12], const T (&f)[6]) {
    grid_inertia_up(&y[0], R, p, I); grid_coriolis_up(&y[10], R, p, B); grid_force_up(&y[22], R, p, f);
}
"""


def gen_branch_frame_library(self):
    for line in _BRANCH_LIBRARY.strip("\n").split("\n"):
        self.gen_add_code_line(line)
    self.gen_add_code_line("")


# M[i][k] = S_i . t1m with the linear part of S_i taken about the reference point of column k's family: about pB it is lin - pB x w
_MKJ_FAM = ("(Spi[0]*t1m[0] + Spi[1]*t1m[1] + Spi[2]*t1m[2]"
            " + (fam ? Spi[3] - (pB[1]*Spi[2] - pB[2]*Spi[1]) : Spi[3])*t1m[3]"
            " + (fam ? Spi[4] - (pB[2]*Spi[0] - pB[0]*Spi[2]) : Spi[4])*t1m[4]"
            " + (fam ? Spi[5] - (pB[0]*Spi[1] - pB[1]*Spi[0]) : Spi[5])*t1m[5])")


# owner walk: S_i about pB dotted with t1m = S_i (about the frame origin) dotted with t1m moved from pB to the origin: n + pB x f - once instead of a cross product per path step
_FAM_FOLD = "if (fam) { w1m[0] += pB[1]*w1m[5] - pB[2]*w1m[4]; w1m[1] += pB[2]*w1m[3] - pB[0]*w1m[5]; w1m[2] += pB[0]*w1m[4] - pB[1]*w1m[3]; }"


def _t1m_lines():
    return ["T t1m[6]; // the I^C S this lane's column of M is built from: about pB for the base half of a long leaf branch, about the branch tip otherwise",
            "{ T x[6], y[6];",
            "  x[0] = S[0]; x[1] = S[1]; x[2] = S[2];",
            "  { const T e0 = myp[0] - pB[0], e1 = myp[1] - pB[1], e2 = myp[2] - pB[2]; x[3] = e1*S[2] - e2*S[1]; x[4] = e2*S[0] - e0*S[2]; x[5] = e0*S[1] - e1*S[0]; }",
            "  grid_rbi_mul(y, ICB, x);",
            "  #pragma unroll",
            "  for (int r = 0; r < 6; r++) { t1m[r] = fam ? y[r] : t1[r]; } }"]


def _anc_local(sig):
    """Strict ancestors (local indices, ascending) of every joint of a component with parent signature sig."""
    out = []
    for k in range(len(sig)):
        a, p = [], sig[k]
        while p >= 0:
            a.append(p)
            p = sig[p]
        out.append(sorted(a))
    return out


def gen_forward_dynamics_gradient_inner_branch(self, use_thread_group=False):
    """The fused inner of the branch-frame path (u-input form): see the module notes."""
    _emit_branch_inner(self, "fdgrad", use_thread_group)


def gen_forward_dynamics_gradient_inner_branch_stream(self, use_thread_group=False):
    """The same inner for the forward_dynamics_gradient KERNEL where the LDS capacity of a CU bounds the resident waves (tuning stream_out,
    decided in helpers/_topology_helpers.gen_lds_layout): one half of the record in LDS at a time, both halves stored from inside."""
    _emit_branch_inner(self, "fdgrad_stream", use_thread_group)


def gen_branch_frame_components(self, use_thread_group=False):
    """Stand-alone kernels of branched robots on the same path: subsets of the fused inner."""
    for mode in ("id", "idgrad", "fd", "minv"):
        _emit_branch_inner(self, mode, use_thread_group)


_BRANCH_MODES = {
    # mode: (function name, what it computes, leading parameters, their docs)
    "fdgrad": ("forward_dynamics_gradient_inner_branch", "Computes the gradient of forward dynamics",
               "T *s_df_du, const T *s_qd, const T *s_u, T *s_X, T *s_SP, T *s_qdd, const robotModel<T> *d_robotModel, const T gravity, const int lane"),
    "fdgrad_stream": ("forward_dynamics_gradient_inner_branch_stream", "Computes the gradient of forward dynamics with ONE half of the record staged in LDS at a time and stores both halves to global memory",
                      "T *d_df_du_k, T *s_df_du, T *s_Y, const T *s_qd, const T *s_u, T *s_X, T *s_SP, T *s_qdd, const robotModel<T> *d_robotModel, const T gravity, const int lane"),
    "id": ("inverse_dynamics_inner_branch", "Compute the RNEA (Recursive Newton-Euler Algorithm)",
           "T *s_c, const T *s_qd, const T *s_qddin, T *s_X, T *s_SP, const robotModel<T> *d_robotModel, const T gravity, const int lane"),
    "idgrad": ("inverse_dynamics_gradient_inner_branch", "Computes the gradient of inverse dynamics",
               "T *s_dc_du, const T *s_qd, const T *s_qddin, T *s_X, T *s_SP, const robotModel<T> *d_robotModel, const T gravity, const int lane"),
    "fd": ("forward_dynamics_inner_branch", "Computes forward dynamics",
           "T *s_qdd, const T *s_qd, const T *s_u, T *s_X, T *s_SP, const robotModel<T> *d_robotModel, const T gravity, const int lane"),
    "minv": ("direct_minv_inner_branch", "Compute the inverse of the mass matrix (dense, symmetric) into LDS",
             "T *s_Minv, T *s_X, T *s_SP, const robotModel<T> *d_robotModel, const int lane"),
}


def _emit_branch_inner(self, mode, use_thread_group=False):
    """One emitter for the five inners of the branch-frame path; `mode` selects the stages (see gen_branch_frame_components)."""
    # "fdgrad_stream" (the forward-dynamics-gradient KERNEL of LDS-capacity-bound robots, tuning stream_out): the image holds ONE half of the result
    # (n^2 values) at a time - pass 1 assembles dc/dqd there, every lane parks its own column (component rows only: n x NCmax values per solve),
    # pass 2 overwrites the same entries with dc/dq; both columns are solved together at the end and leave one half after the other: the
    # staging shrinks from 2 n^2 to n^2 + n NCmax values per solve = more resident waves per CU
    stream = mode == "fdgrad_stream"
    fname, fdoc, fsig = _BRANCH_MODES[mode]
    if stream:
        mode = "fdgrad"
    grad = mode in ("fdgrad", "idgrad")      # needs the Coriolis composites and the derivative entries
    kin = mode != "minv"                      # needs velocities / accelerations
    needs_M = mode in ("fdgrad", "fd", "minv")
    qdd_in = mode in ("id", "idgrad")        # joint accelerations are an input (id: the pointer may be null = zero)
    m = self.model
    n = m.n
    P = self.branch_plan
    D, maxLb, maxlevel, maxchild, RL = P["D"], P["maxLb"], P["maxlevel"], P["maxchild"], P["row_len"]
    lanes = self.lanes_per_solve
    ld = self.minv_ld
    tab = self.branch_tab_offset
    H = 24
    A = self.gen_add_code_line
    outdoc = {"fdgrad": "s_df_du receives -Minv*dc/du in the device layout [col*n + row] (2*NUM_JOINTS*NUM_JOINTS values; also the assembly area of dc/du)" if not stream else
              "d_df_du_k is this solve's record in global memory (2*NUM_JOINTS*NUM_JOINTS values, layout [col*n + row]; nullptr: nothing is stored); s_df_du is LDS for ONE half of it (NUM_JOINTS*NUM_JOINTS values); s_Y is LDS for the parked dc/dqd columns (NUM_JOINTS*%d values)" % max(len(s_) for s_ in P["shapes"]),
              "id": "s_c receives the joint torques (lane of joint j writes s_c[j]); s_qddin may be nullptr (zero accelerations)",
              "idgrad": "s_dc_du receives dc/du in the device layout [col*n + row], col in [0,2n) = [d/dq | d/dqd]",
              "fd": "s_qdd receives the joint accelerations (it also holds tau - c on the way)",
              "minv": "s_Minv receives the dense symmetric inverse of the joint-space inertia (leading dimension GRID_MINV_LD; zero between base-rooted components)"}[mode]
    self.gen_add_func_doc(fdoc + ", every branch of the tree in the frame of its tip link",
                          ["robots whose joints are all revolute (see the module notes of algorithms/_branch_frame_gradient.py)", outdoc,
                           "the caller must grid_wave_sync() before other lanes read the result"],
                          ["s_X is this solve's compact X(q) storage (the rotation blocks are read; once the frames are known it is re-used for the",
                           "     tree-sparse M, its factors and the branch hand-over records)",
                           "s_SP is LDS scratch for the joint axes along the root path of every branch (6 values per path joint, plus one spare set of records per solve where the kernel has no result image to take them%s)" % ("" if all(v[0] == "x" for v in P["place"].values()) else "; then what does not fit into s_X: " + ", ".join(k for k, v in P["place"].items() if v[0] == "sp")),
                           "d_robotModel is the pointer to the initialized model specific helpers on the GPU",
                           "lane is the caller's lane index inside the solve's lane group"], None)
    A("template <typename T>")
    A("__device__ __forceinline__")
    A("void %s(%s) {" % (fname, fsig), True)
    ts_mode = mode == "fdgrad" and not stream and self.tuning["debug_stop"] == 20  # profiling build: per-wave cycle stamps at the phase boundaries replace the first outputs

    def TS(i):
        if ts_mode:
            A("asm volatile(\"s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_memtime %%0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(grid_ts[%d]) : : \"memory\");" % i)

    if ts_mode:
        A("unsigned long long grid_ts[10];")
    TS(0)
    A("const T Z = static_cast<T>(0);")
    A("const T *d_L = &grid_model_constants(static_cast<const T *>(nullptr))[%d + %d*lane]; // this lane's table row (gen_branch_frame_constants)" % (tab, RL))
    A("(void)d_robotModel;")
    A("T Lc[12]; // link constants: Ic (6, about the centre of mass), c (3), m, damping, axis")
    A("#pragma unroll")
    A("for (int r = 0; r < 12; r++) { Lc[r] = d_L[r]; }")
    A("const int jid = static_cast<int>(d_L[12]); const bool active = jid >= 0; const int js = active ? jid : 0;")
    A("const int pos = static_cast<int>(d_L[13]), Lb = static_cast<int>(d_L[14]), level = static_cast<int>(d_L[15]), slot = static_cast<int>(d_L[16]);")
    A("const int cbase = static_cast<int>(d_L[17]), shape = static_cast<int>(d_L[18]), plen = static_cast<int>(d_L[19]), ubase = static_cast<int>(d_L[20]), mstart = static_cast<int>(d_L[21]);")
    for c in range(maxchild):
        A("const int cs%d = static_cast<int>(d_L[%d]); // child branch %d of this lane's branch (-1: none)" % (c, H + D + c, c))
    ptr = lambda item: "&s_%s[%d]" % ("X" if P["place"][item][0] == "x" else "SP", P["place"][item][1])
    A("T *s_Mc = %s, *s_Uc = %s + ubase, *s_G = %s, *s_trash = %s; // (s_X: valid once the frame chain is done with X(q))" % (ptr("M"), ptr("U"), ptr("G"), ptr("trash")))
    A("const int own = Lb - 1 - pos; // index of this lane's joint on the root path of its branch (tip -> root); -1 on lanes without a joint")
    bfam = needs_M and any(r_ >= 0 for r_ in self.gen_branch_frame_constants()[22::RL])
    if bfam:
        A("const int iB = static_cast<int>(d_L[22]); const bool fam = d_L[23] > static_cast<T>(0.5); // base-origin family of the joint-space inertia (leaf branches of >= 5 joints)")
    A("const int li = jid - cbase;   // index of the joint inside its base-rooted component")
    owner = P["owner"]
    if owner:
        PL = H + D + max(maxchild, 1) + 3 * D
        for i in range(D):
            A("const int plc%d = static_cast<int>(d_L[%d]); const int pl%d = (plc%d >= 0) ? (plc%d & 63) : lane; const bool cr%d = plc%d >= 64; // lane that owns path joint %d; the path enters the parent branch here"
              % (i, PL + i, i, i, i, i, i, i))
        A("const int lparc = static_cast<int>(d_L[%d]); const int lpar = (lparc >= 0) ? lparc : lane; // lane of the link this lane's branch hangs off" % (RL - 3))
        A("(void)s_SP; (void)lpar;" + "".join(" (void)pl%d; (void)cr%d;" % (i, i) for i in range(D)))
    else:
        A("T *s_Sp = &s_SP[%d*slot]; // joint axes along the root path of this lane's branch, in the branch frame" % (6 * D))
    A("(void)level; (void)plen; (void)s_G; (void)mstart; (void)s_Mc; (void)s_Uc; (void)s_trash; (void)shape; (void)li; (void)ubase;")
    for i in range(D):
        A("const int pc%d = static_cast<int>(d_L[%d]); const bool pv%d = pc%d >= 0; const int pj%d = pv%d ? (pc%d >> 2) : js; const bool act%d = pv%d && (%d >= own);"
          % (i, H + i, i, i, i, i, i, i, i, i))
    zero = {"fdgrad": ("s_df_du", n * n if stream else 2 * n * n), "idgrad": ("s_dc_du", 2 * n * n), "minv": ("s_Minv", n * ld)}.get(mode)
    A("grid_wave_sync();")
    spare_in_image = zero is not None and branch_spare_in_image(self, zero[1])
    lanes_head = (6 * D + 3) // 4 if (spare_in_image and not branch_chain_scan(self) and not owner) else 0  # quads at the head of the image that take the spare path records of the frame chain
    if zero is not None:
        A("// zero image of the result (unrelated joints, and rows outside the component of a column, stay exactly zero)")
        A("for (int e = lane + %d; e < %d; e += %d) {" % (lanes_head, zero[1] // 4, lanes), True)
        A("#pragma unroll")
        A("for (int r = 0; r < 4; r++) { %s[4*e + r] = Z; }" % zero[0])
        self.gen_add_end_control_flow()
        if stream and owner:
            ylen = (n * max(len(s_) for s_ in P["shapes"]) + 3) // 4
            A("for (int e = lane; e < %d; e += %d) { // ... and of the parked d/dqd columns (pass 2 writes their entries directly)" % (ylen, lanes), True)
            A("#pragma unroll")
            A("for (int r = 0; r < 4; r++) { s_Y[4*e + r] = Z; }")
            self.gen_add_end_control_flow()
        if zero[1] % 4:  # (never past the end of the image: the next lane group's image starts there)
            A("if (lane < %d) { %s[%d + lane] = Z; }" % (zero[1] % 4, zero[0], zero[1] // 4 * 4))
    # ------------------------------------------------------------------ frame chain along the root path
    TS(1)
    A("//")
    A("// frames: walk the root path of this lane's branch, tip -> root")
    A("//")
    A("T myR[9], myp[3], TR[9], Tp[3], gvec[3] = {Z, Z, Z};")
    if bfam:
        A("T pB[3] = {Z, Z, Z}; // second reference point: the frame origin of the branch's middle joint, in the branch frame")
    chain_scan = owner or branch_chain_scan(self)
    if chain_scan:
        _emit_chain_scan(self, P, H, RL, bfam, kin, use_thread_group, ptr, owner)
    else:
        A("{", True)
        ro = H + D + max(maxchild, 1)
        A("T rj[%d][3]; // origins of the path joints' frames in their parents' coordinates: all table reads issued before the first use" % D)
        for i in range(D - 1):
            A("rj[%d][0] = d_L[%d]; rj[%d][1] = d_L[%d]; rj[%d][2] = d_L[%d];" % (i, ro + 3 * i, i, ro + 3 * i + 1, i, ro + 3 * i + 2))
        A("T Rc[9] = {static_cast<T>(1), Z, Z, Z, static_cast<T>(1), Z, Z, Z, static_cast<T>(1)};")
        A("T pc[3] = {Z, Z, Z};")
        A("#pragma unroll")
        A("for (int r = 0; r < 9; r++) { myR[r] = Rc[r]; TR[r] = Rc[r]; }")
        A("#pragma unroll")
        A("for (int r = 0; r < 3; r++) { myp[r] = pc[r]; Tp[r] = pc[r]; }")
        junction_steps = sorted(set(len(P["branches"][b]) for b in range(P["nb"]) if P["pb"][b] >= 0))
        root_steps = sorted(set(len(p_) - 1 for p_ in P["paths"]))
        A("T En[9]; // E(q) of the next path joint (parent -> child coordinates, row-major): read one step ahead")
        A("#pragma unroll")
        A("for (int r = 0; r < 9; r++) { En[r] = s_X[GRID_X_STRIDE*pj0 + r]; }")
        A("T *s_sp_dst = (active && pos == 0) ? s_Sp : %s; // the first lane of every branch parks the joint axes of the path; the others write to %s"
          % ((zero[0], "the head of the result image (cleared below)") if spare_in_image else ("&s_SP[%d]" % P["sp_spare"], "a spare set of records")))
        for i in range(D):
            A("{ // path step %d" % i, True)
            A("T Ei[9];")
            A("#pragma unroll")
            A("for (int r = 0; r < 9; r++) { Ei[r] = En[r]; }")
            if i < D - 1:
                A("#pragma unroll")
                A("for (int r = 0; r < 9; r++) { En[r] = s_X[GRID_X_STRIDE*pj%d + r]; }" % (i + 1))
            if 0 < i < maxLb:
                A("#pragma unroll")
                A("for (int r = 0; r < 9; r++) { myR[r] = (own == %d) ? Rc[r] : myR[r]; }" % i)
                A("#pragma unroll")
                A("for (int r = 0; r < 3; r++) { myp[r] = (own == %d) ? pc[r] : myp[r]; }" % i)
            if bfam and 0 < i < maxLb:
                A("pB[0] = (iB == %d) ? pc[0] : pB[0]; pB[1] = (iB == %d) ? pc[1] : pB[1]; pB[2] = (iB == %d) ? pc[2] : pB[2];" % (i, i, i))
            if i in junction_steps:
                A("#pragma unroll")
                A("for (int r = 0; r < 9; r++) { TR[r] = (Lb == %d) ? Rc[r] : TR[r]; } // pose of the parent branch's frame in this branch's frame" % i)
                A("#pragma unroll")
                A("for (int r = 0; r < 3; r++) { Tp[r] = (Lb == %d) ? pc[r] : Tp[r]; }" % i)
            A("{ // joint axis of path joint %d in the branch frame, parked for the walks below (zero beyond the root)" % i)
            A("  const int pa = pc%d & 3; T w[3];" % i)
            A("  #pragma unroll")
            A("  for (int r = 0; r < 3; r++) { const T wr = (pa == 0) ? Rc[3*r] : ((pa == 1) ? Rc[3*r + 1] : Rc[3*r + 2]); w[r] = pv%d ? wr : Z; }" % i)
            A("  s_sp_dst[%d] = w[0]; s_sp_dst[%d] = w[1]; s_sp_dst[%d] = w[2];" % (6 * i, 6 * i + 1, 6 * i + 2))
            A("  s_sp_dst[%d] = pc[1]*w[2] - pc[2]*w[1]; s_sp_dst[%d] = pc[2]*w[0] - pc[0]*w[2]; s_sp_dst[%d] = pc[0]*w[1] - pc[1]*w[0]; }" % (6 * i + 3, 6 * i + 4, 6 * i + 5))
            A("T Rn[9];")
            A("#pragma unroll")
            A("for (int r = 0; r < 3; r++) {", True)
            A("#pragma unroll")
            A("for (int c = 0; c < 3; c++) { Rn[3*r + c] = Rc[3*r]*Ei[c] + Rc[3*r + 1]*Ei[3 + c] + Rc[3*r + 2]*Ei[6 + c]; }")
            self.gen_add_end_control_flow()
            if i in root_steps and kin:
                A("gvec[0] = (plen == %d) ? gravity*Rn[2] : gvec[0]; gvec[1] = (plen == %d) ? gravity*Rn[5] : gvec[1]; gvec[2] = (plen == %d) ? gravity*Rn[8] : gvec[2]; // base acceleration (0,0,g) in this branch's coordinates" % (i + 1, i + 1, i + 1))
            if i < D - 1:
                A("#pragma unroll")
                A("for (int r = 0; r < 3; r++) { pc[r] -= Rn[3*r]*rj[%d][0] + Rn[3*r + 1]*rj[%d][1] + Rn[3*r + 2]*rj[%d][2]; }" % (i, i, i))
                A("#pragma unroll")
                A("for (int r = 0; r < 9; r++) { Rc[r] = Rn[r]; }")
                A("GRID_SCHED_FENCE();")
            self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
    _probe(self, "chain", "myR:9", "myp:3", "TR:9", "Tp:3", "gvec:3")
    self.gen_add_sync(use_thread_group)
    if spare_in_image and not chain_scan:
        A("// (the head of the result image took the spare path records: clear it now)")
        A("if (lane < %d) {" % lanes_head, True)
        A("#pragma unroll")
        A("for (int r = 0; r < 4; r++) { %s[4*lane + r] = Z; }" % zero[0])
        self.gen_add_end_control_flow()
    # ------------------------------------------------------------------ own link: S, inertia in F_b
    TS(2)
    A("//")
    A("// this lane's link in its branch frame: joint axis, inertia; velocity and bias acceleration by the running sums along the path")
    A("//")
    steps = [k for k in (1, 2, 4, 8) if k < maxLb]
    A("T mkd[GRID_SCAN_STEPS];")
    if not steps:
        A("mkd[0] = Z;")
    for s_, k in enumerate(steps):
        A("mkd[%d] = (pos + %d < Lb) ? static_cast<T>(1) : Z;" % (s_, k))
    if kin:
        A("const T qd = active ? s_qd[js] : Z;")
    else:
        A("(void)gvec;")
    A("T S[6];")
    A("{ const int ax = static_cast<int>(Lc[11]);")
    A("  #pragma unroll")
    A("  for (int r = 0; r < 3; r++) { S[r] = (ax == 0) ? myR[3*r] : ((ax == 1) ? myR[3*r+1] : myR[3*r+2]); } }")
    A("S[3] = myp[1]*S[2] - myp[2]*S[1]; S[4] = myp[2]*S[0] - myp[0]*S[2]; S[5] = myp[0]*S[1] - myp[1]*S[0];")
    A("T I[10]; // this link's inertia about the origin of its branch frame")
    if bfam:
        A("T IB[10]; // ... and about pB")
    A("{", True)
    A("T d[3], RI[9];")
    A("#pragma unroll")
    A("for (int r = 0; r < 3; r++) {", True)
    A("d[r] = myp[r] + myR[3*r]*Lc[6] + myR[3*r+1]*Lc[7] + myR[3*r+2]*Lc[8]; // centre of mass")
    A("RI[3*r]   = myR[3*r]*Lc[0] + myR[3*r+1]*Lc[1] + myR[3*r+2]*Lc[2];")
    A("RI[3*r+1] = myR[3*r]*Lc[1] + myR[3*r+1]*Lc[3] + myR[3*r+2]*Lc[4];")
    A("RI[3*r+2] = myR[3*r]*Lc[2] + myR[3*r+1]*Lc[4] + myR[3*r+2]*Lc[5];")
    self.gen_add_end_control_flow()
    A("T rot[6];")
    pairs = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))
    for k, (r_, c_) in enumerate(pairs):
        A("rot[%d] = RI[%d]*myR[%d] + RI[%d]*myR[%d] + RI[%d]*myR[%d];" % (k, 3 * r_, 3 * c_, 3 * r_ + 1, 3 * c_ + 1, 3 * r_ + 2, 3 * c_ + 2))

    def parallel_axis(dst, dv):
        A("{ const T md0 = Lc[9]*%s[0], md1 = Lc[9]*%s[1], md2 = Lc[9]*%s[2];" % (dv, dv, dv))
        for k, (r_, c_) in enumerate(pairs):
            if r_ == c_:
                o1, o2 = [x for x in range(3) if x != r_]
                A("  %s[%d] = rot[%d] + md%d*%s[%d] + md%d*%s[%d];" % (dst, k, k, o1, dv, o1, o2, dv, o2))
            else:
                A("  %s[%d] = rot[%d] - md%d*%s[%d];" % (dst, k, k, r_, dv, c_))
        A("  %s[6] = md0; %s[7] = md1; %s[8] = md2; %s[9] = Lc[9]; }" % (dst, dst, dst, dst))

    parallel_axis("I", "d")
    if bfam:
        A("T dB[3] = {d[0] - pB[0], d[1] - pB[1], d[2] - pB[2]}; // centre of mass relative to pB")
        parallel_axis("IB", "dB")
    self.gen_add_end_control_flow()
    A("const T damping = Lc[10]; (void)damping;")
    _probe(self, "link", "S:6", "I:10")
    if bfam:
        _probe(self, "link", "IB:10")

    def walk_open(with_qdd):
        """Software-pipelined walk along the root path, root -> tip: the LDS reads of step i-1 are issued before the arithmetic of step i and
        a scheduling fence closes every step (otherwise the compiler hoists all the loads of the walk to its top and spills)."""
        A("T Sn[6]%s%s;" % (", qdn" if kin else "", ", qddn" if with_qdd else ""))
        walk_fetch(D - 1, with_qdd)

    def walk_fetch(i, with_qdd):
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { Sn[r] = s_Sp[%d + r]; }" % (6 * i))
        if kin:
            A("qdn = act%d ? s_qd[pj%d] : Z;" % (i, i))
        if with_qdd:
            if mode == "id":
                A("qddn = (act%d && s_qddin != nullptr) ? s_qddin[pj%d] : Z;" % (i, i))
            else:
                A("qddn = act%d ? %s[pj%d] : Z;" % (i, "s_qddin" if qdd_in else "s_qdd", i))

    def walk_step(i, with_qdd):
        A("{ // path step %d" % i, True)
        A("T Spi[6];%s%s" % (" const T qdi = qdn;" if kin else "", " const T qddi = qddn;" if with_qdd else ""))
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { Spi[r] = Sn[r]; }")
        if i > 0:
            walk_fetch(i - 1, with_qdd)

    def walk_close():
        A("GRID_SCHED_FENCE();")
        self.gen_add_end_control_flow()

    def hand_down(name, what):
        """owner walk: the lanes of tree level lv add the parent link's `name` (final after level lv - 1), re-expressed in their branch frame"""
        for lv in range(1, maxlevel + 1):
            A("{ // tree level %d: %s of the link the branch hangs off, in this branch's frame" % (lv, what), True)
            A("T xi[6], y[6];")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { xi[r] = grid_group_shfl(%s[r], lpar); }" % name)
            A("grid_motion_down(y, TR, Tp, xi);")
            A("if (level == %d) {" % lv, True)
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { %s[r] += y[r]; }" % name)
            self.gen_add_end_control_flow()
            self.gen_add_end_control_flow()

    cross_steps = sorted(set(i for b in range(P["nb"]) for i in range(1, len(P["paths"][b])) if P["br"][P["paths"][b][i]] != P["br"][P["paths"][b][i - 1]]))

    def owner_walk(fetch, fvecs, cvecs, body):
        """owner walk along the root path, tip -> root: per step the owner's vectors (`fetch`: S, Pd, Pdd of path joint i in the frame of ITS branch) arrive by
        cross-lane reads, one step ahead of their use; where the path enters the parent branch (cr_i) the lane's force-type vectors `fvecs` and the couples
        `cvecs` (3 values, no linear part) are re-expressed in the parent branch's frame."""
        loc = {"S": "Spi", "Pd": "Pdi", "Pdd": "Pddi"}
        nxt = {"S": "Sn", "Pd": "Pdn", "Pdd": "Pddn"}
        A("T " + ", ".join("%s[6]" % nxt[f] for f in fetch) + ";")

        def fetch_step(i):
            for f in fetch:
                A("#pragma unroll")
                A("for (int r = 0; r < 6; r++) { %s[r] = grid_group_shfl(%s[r], pl%d); }" % (nxt[f], f, i))

        fetch_step(0)
        for i in range(D):
            if i in cross_steps:
                A("{ // the root paths of some branches enter their parent branch at step %d: their vectors move into that branch's frame" % i, True)
                if maxlevel > 1:
                    A("T R_[9], p_[3]; // pose of the next frame in the current one: TR, Tp of the lanes of the branch that ends here")
                    A("#pragma unroll")
                    A("for (int r = 0; r < 9; r++) { R_[r] = grid_group_shfl(TR[r], pl%d); }" % (i - 1))
                    A("#pragma unroll")
                    A("for (int r = 0; r < 3; r++) { p_[r] = grid_group_shfl(Tp[r], pl%d); }" % (i - 1))
                else:
                    A("const T (&R_)[9] = TR; const T (&p_)[3] = Tp;")
                A("if (cr%d) {" % i, True)
                for f in fvecs:
                    A("{ T y[6]; grid_force_up(y, R_, p_, %s);" % f)
                    A("  #pragma unroll")
                    A("  for (int r = 0; r < 6; r++) { %s[r] = y[r]; } }" % f)
                for c in cvecs:
                    A("{ T y[3]; grid_rt3(y, R_, %s[0], %s[1], %s[2]); %s[0] = y[0]; %s[1] = y[1]; %s[2] = y[2]; }" % (c, c, c, c, c, c))
                self.gen_add_end_control_flow()
                self.gen_add_end_control_flow()
            A("{ // path step %d" % i, True)
            A("T " + ", ".join("%s[6]" % loc[f] for f in fetch) + ";")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { " + " ".join("%s[r] = %s[r];" % (loc[f], nxt[f]) for f in fetch) + " }")
            if i < D - 1:
                fetch_step(i + 1)
            body(i)
            A("GRID_SCHED_FENCE();")
            self.gen_add_end_control_flow()

    if kin and owner:
        A("T mku[GRID_SCAN_STEPS]; // prefix masks: the partner lane of step s is a joint of the same branch")
        if not steps:
            A("mku[0] = Z;")
        for s_, k in enumerate(steps):
            A("mku[%d] = (pos >= %d) ? static_cast<T>(1) : Z;" % (s_, k))
        A("T v[6], a[6];")
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { v[r] = S[r]*qd; }")
        A("grid_prefix_sum(v, mku); // joints of the own branch up to this one")
        hand_down("v", "velocity")
        A("T Pd[6]; grid_mxm(Pd, v, S); // = S-dot of the own joint")
        if qdd_in:
            A("const T qddo = %s;" % ("(active && s_qddin != nullptr) ? s_qddin[js] : Z" if mode == "id" else "active ? s_qddin[js] : Z"))
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { a[r] = Pd[r]*qd%s; }" % (" + S[r]*qddo" if qdd_in else ""))
        A("grid_prefix_sum(a, mku);")
        A("a[3] += gvec[0]; a[4] += gvec[1]; a[5] += gvec[2]; // (zero below tree level 0)")
        hand_down("a", "acceleration")
        _probe(self, "vel", "v:6", "a:6", "Pd:6")
    elif kin:
        A("T v[6] = {Z, Z, Z, Z, Z, Z}, a[6] = {Z, Z, Z, Z, Z, Z};")
        A("{", True)
        walk_open(qdd_in)
        for i in range(D - 1, -1, -1):
            walk_step(i, qdd_in)
            A("T Pdi[6];")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { v[r] += Spi[r]*qdi; }")
            A("grid_mxm(Pdi, v, Spi);")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { a[r] += Pdi[r]*qdi%s; }" % (" + Spi[r]*qddi" if qdd_in else ""))
            walk_close()
        self.gen_add_end_control_flow()
        A("a[3] += gvec[0]; a[4] += gvec[1]; a[5] += gvec[2];")
        A("T Pd[6]; grid_mxm(Pd, v, S); // = S-dot of the own joint")
    # body quantities and composites (as _tip_frame_gradient._emit_bias, with a given)
    TS(3)
    A("T IC[10], BC[12], fC[6];")
    if grad:
        A("{", True)
        A("T Iv[6]; grid_rbi_mul(Iv, I, v); // [n; l]: the link's momentum")
        A("grid_rbi_mul(fC, I, a); grid_fxv_peq(fC, v, Iv);")
        A("const T A00 = v[1]*I[2] - v[2]*I[1], A01 = v[1]*I[4] - v[2]*I[3], A02 = v[1]*I[5] - v[2]*I[4];")
        A("const T A10 = v[2]*I[0] - v[0]*I[2], A11 = v[2]*I[1] - v[0]*I[4], A12 = v[2]*I[2] - v[0]*I[5];")
        A("const T A20 = v[0]*I[1] - v[1]*I[0], A21 = v[0]*I[3] - v[1]*I[1], A22 = v[0]*I[4] - v[1]*I[2];")
        A("const T uh = static_cast<T>(2)*(v[3]*I[6] + v[4]*I[7] + v[5]*I[8]);")
        A("BC[0] = static_cast<T>(2)*(A00 - I[6]*v[3]) + uh;")
        A("BC[1] = A01 + A10 - I[6]*v[4] - I[7]*v[3];")
        A("BC[2] = A02 + A20 - I[6]*v[5] - I[8]*v[3];")
        A("BC[3] = static_cast<T>(2)*(A11 - I[7]*v[4]) + uh;")
        A("BC[4] = A12 + A21 - I[7]*v[5] - I[8]*v[4];")
        A("BC[5] = static_cast<T>(2)*(A22 - I[8]*v[5]) + uh;")
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { BC[6 + r] = Iv[r]; }")
        self.gen_add_end_control_flow()
    elif kin:
        A("{ T Iv[6]; grid_rbi_mul(Iv, I, v); grid_rbi_mul(fC, I, a); grid_fxv_peq(fC, v, Iv); } // the link's force; no Coriolis matrix needed")
    use_I = mode != "id"   # composite inertia
    use_B = grad            # composite Coriolis matrix
    use_f = kin             # composite force
    if not use_B:
        A("(void)BC;")
    if not use_f:
        A("(void)fC;")
    if use_I:
        A("#pragma unroll")
        A("for (int r = 0; r < 10; r++) { IC[r] = I[r]; }")
        A("grid_suffix_sum(IC, mkd);")
        if bfam:
            A("T ICB[10]; // composite about pB (leaf branches only: nothing is handed up into it)")
            A("#pragma unroll")
            A("for (int r = 0; r < 10; r++) { ICB[r] = IB[r]; }")
            A("grid_suffix_sum(ICB, mkd);")
    else:
        A("(void)IC;")
    if use_B:
        A("grid_suffix_sum(BC, mkd);")
    if use_f:
        A("grid_suffix_sum(fC, mkd);")
    A("// (composites over the rest of the branch so far)")
    if use_I:
        _probe(self, "comp_scan", "IC:10")
        if bfam:
            _probe(self, "comp_scan", "ICB:10")
    for lv in range(maxlevel, 0, -1):
        A("{ // tree level %d -> %d: branch totals re-expressed in the parent branch's frame and handed to all of its lanes" % (lv, lv - 1), True)
        A("T y[28];")
        if use_I:
            A("grid_inertia_up(&y[0], TR, Tp, IC);")
        if use_B:
            A("grid_coriolis_up(&y[10], TR, Tp, BC);")
        if use_f:
            A("grid_force_up(&y[22], TR, Tp, fC);")
        if use_I:
            _probe(self, "handover", "y:10")
        A("if (active && level == %d && pos == 0) {" % lv, True)
        for (u_, lo_, hi_) in ((use_I, 0, 10), (use_B, 10, 22), (use_f, 22, 28)):
            if u_:
                A("#pragma unroll")
                A("for (int r = %d; r < %d; r++) { s_G[28*slot + r] = y[r]; }" % (lo_, hi_))
        self.gen_add_end_control_flow()
        self.gen_add_sync(use_thread_group)
        for c in range(maxchild):
            A("{ const int cs = cs%d;" % c)
            A("  if (level == %d && cs >= 0) {" % (lv - 1), True)
            A("const T *g = &s_G[28*cs];")
            if use_I:
                A("#pragma unroll")
                A("for (int r = 0; r < 10; r++) { IC[r] += g[r]; }")
            if use_B:
                A("#pragma unroll")
                A("for (int r = 0; r < 12; r++) { BC[r] += g[10 + r]; }")
            if use_f:
                A("#pragma unroll")
                A("for (int r = 0; r < 6; r++) { fC[r] += g[22 + r]; }")
            self.gen_add_end_control_flow()
            A("}")
        self.gen_add_end_control_flow()
    if mode == "id":
        A("if (active) { s_c[jid] = grid_dot6(S, fC) + damping*qd; }")
        self.gen_add_end_function()
        return
    if mode == "idgrad":
        # accelerations are an input: one walk produces every entry of dc/du that couples this joint with its ancestors
        A("T t1[6], t2[6], t3[6], t4[3], Pdd[6];")
        A("grid_mxm(Pdd, a, S); grid_mxm_peq(Pdd, v, Pd);")
        A("grid_rbi_mul(t1, IC, S);")
        A("grid_bmul(t2, BC, S); grid_rbi_mul_peq(t2, IC, Pd, static_cast<T>(2));")
        A("grid_bmul(t3, BC, Pd); grid_rbi_mul_peq(t3, IC, Pdd, static_cast<T>(1)); grid_fxv_peq(t3, S, fC);")
        A("grid_btmul(t4, BC, S);")
        if owner:
            A("{", True)

            def body(i):
                A("const T up_q = grid_dot6(Spi, t3), up_d = grid_dot6(Spi, t2);")
                A("const T lo_q = grid_dot6(t1, Pddi) + t4[0]*Pdi[0] + t4[1]*Pdi[1] + t4[2]*Pdi[2];")
                A("const T lo_d = static_cast<T>(2)*grid_dot6(t1, Pdi) + t4[0]*Spi[0] + t4[1]*Spi[1] + t4[2]*Spi[2];")
                A("*(act%d ? &s_dc_du[jid*%d + pj%d] : s_trash) = up_q;" % (i, n, i))
                A("*(act%d ? &s_dc_du[(%d + jid)*%d + pj%d] : s_trash) = up_d + ((own == %d) ? damping : Z); // + damping on the diagonal (oracle _test.py:486)" % (i, n, n, i, i))
                A("*((act%d && own != %d) ? &s_dc_du[pj%d*%d + jid] : s_trash) = lo_q;" % (i, i, i, n))
                A("*((act%d && own != %d) ? &s_dc_du[(%d + pj%d)*%d + jid] : s_trash) = lo_d;" % (i, i, n, i, n))

            owner_walk(("S", "Pd", "Pdd"), ("t1", "t2", "t3"), ("t4",), body)
            self.gen_add_end_control_flow()
            self.gen_add_end_function()
            return
        A("{", True)
        A("T vr[6] = {Z, Z, Z, Z, Z, Z}, ar[6] = {Z, Z, Z, gvec[0], gvec[1], gvec[2]};")
        walk_open(True)
        for i in range(D - 1, -1, -1):
            walk_step(i, True)
            A("T Pdi[6], Pddi[6];")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { vr[r] += Spi[r]*qdi; }")
            A("grid_mxm(Pdi, vr, Spi);")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { ar[r] += Pdi[r]*qdi + Spi[r]*qddi; }")
            A("grid_mxm(Pddi, ar, Spi); grid_mxm_peq(Pddi, vr, Pdi);")
            A("const T up_q = grid_dot6(Spi, t3), up_d = grid_dot6(Spi, t2);")
            A("const T lo_q = grid_dot6(t1, Pddi) + t4[0]*Pdi[0] + t4[1]*Pdi[1] + t4[2]*Pdi[2];")
            A("const T lo_d = static_cast<T>(2)*grid_dot6(t1, Pdi) + t4[0]*Spi[0] + t4[1]*Spi[1] + t4[2]*Spi[2];")
            A("*(act%d ? &s_dc_du[jid*%d + pj%d] : s_trash) = up_q;" % (i, n, i))
            A("*(act%d ? &s_dc_du[(%d + jid)*%d + pj%d] : s_trash) = up_d + ((own == %d) ? damping : Z); // + damping on the diagonal (oracle _test.py:486)" % (i, n, n, i, i))
            A("*((act%d && own != %d) ? &s_dc_du[pj%d*%d + jid] : s_trash) = lo_q;" % (i, i, i, n))
            A("*((act%d && own != %d) ? &s_dc_du[(%d + pj%d)*%d + jid] : s_trash) = lo_d;" % (i, i, n, i, n))
            walk_close()
        self.gen_add_end_control_flow()
        self.gen_add_end_function()
        return
    if use_I:
        _probe(self, "comp_I", "IC:10")
    if use_B:
        _probe(self, "comp_BF", "BC:12", "fC:6")
    # ------------------------------------------------------------------ pass 1
    TS(4)
    yo = 0 if stream else n  # column offset of the d/dqd half inside the image
    if mode == "fdgrad":
        A("// everything that does not depend on qdd: t1, t2, t4, tau - c; then the entries that couple this joint with its ancestors")
        A("T t1[6], t4[3];")
        A("grid_rbi_mul(t1, IC, S);")
        A("grid_btmul(t4, BC, S);")
        if bfam:
            for ln in _t1m_lines():
                A(ln)
            _probe(self, "t1", "t1m:6")
        _probe(self, "t1", "t1:6")
        A("if (active) { s_qdd[jid] = s_u[jid] - (grid_dot6(S, fC) + damping*qd); }")
        A("{", True)
        if not owner:
            A("T t2[6]; grid_bmul(t2, BC, S); grid_rbi_mul_peq(t2, IC, Pd, static_cast<T>(2));")
        if owner:  # only the joint-space inertia here (the factorisation waits for it); the d/dqd entries join pass 2, which fetches Pd_i anyway
            A("T w1m[6]; // (t1 stays in the branch frame for pass 2)")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { w1m[r] = %s[r]; }" % ("t1m" if bfam else "t1"))
            if bfam:
                A(_FAM_FOLD)
            owner_walk(("S",), ("w1m",), (), lambda i: A("*(act%d ? &s_Mc[mstart + plen - %d] : s_trash) = %s; // (ancestors in ascending order, then the diagonal)"
                       % (i, i + 1, ("static_cast<T>(static_cast<float>(%s))" if "M" in tuple(self.tuning.get("round_probe", ())) else "%s") % "grid_dot6(Spi, w1m)")))
        else:
            A("T vr[6] = {Z, Z, Z, Z, Z, Z};")
            walk_open(False)
        for i in (() if owner else range(D - 1, -1, -1)):
            walk_step(i, False)
            A("T Pdi[6];")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { vr[r] += Spi[r]*qdi; }")
            A("grid_mxm(Pdi, vr, Spi);")
            if bfam:
                A("const T mkj = %s, up_d = grid_dot6(Spi, t2);" % _MKJ_FAM)
            else:
                A("const T mkj = grid_dot6(Spi, t1), up_d = grid_dot6(Spi, t2);")
            A("const T lo_d = static_cast<T>(2)*grid_dot6(t1, Pdi) + t4[0]*Spi[0] + t4[1]*Spi[1] + t4[2]*Spi[2];")
            A("// (branch-free: lanes that have no such entry write to a spare word)")
            A("*(act%d ? &s_Mc[mstart + plen - %d] : s_trash) = %s; // (ancestors in ascending order, then the diagonal)" % (i, i + 1, "static_cast<T>(static_cast<float>(mkj))" if "M" in tuple(self.tuning.get("round_probe", ())) else "mkj"))
            A("*(act%d ? &s_df_du[(%d + jid)*%d + pj%d] : s_trash) = up_d + ((own == %d) ? damping : Z); // + damping on the diagonal (oracle _test.py:486)" % (i, yo, n, i, i))
            A("*((act%d && own != %d) ? &s_df_du[(%d + pj%d)*%d + jid] : s_trash) = lo_d;" % (i, i, yo, i, n))
            walk_close()
        self.gen_add_end_control_flow()
    else:  # fd, minv: only the joint-space inertia, M[i][k] = S_i . (I^C_k S_k) for the ancestors-or-self i of this lane's joint k
        A("T t1[6]; grid_rbi_mul(t1, IC, S);")
        if bfam:
            for ln in _t1m_lines():
                A(ln)
        if mode == "fd":
            A("if (active) { s_qdd[jid] = s_u[jid] - (grid_dot6(S, fC) + damping*qd); }")
        A("{", True)
        if owner:
            A("T (&w1m)[6] = %s;" % ("t1m" if bfam else "t1"))
            if bfam:
                A(_FAM_FOLD)
            owner_walk(("S",), ("w1m",), (), lambda i: A("*(act%d ? &s_Mc[mstart + plen - %d] : s_trash) = grid_dot6(Spi, w1m); // (ancestors in ascending order, then the diagonal)" % (i, i + 1)))
        else:
            A("T Sn[6];")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { Sn[r] = s_Sp[%d + r]; }" % (6 * (D - 1)))
        for i in (() if owner else range(D - 1, -1, -1)):
            A("{ T Spi[6];")
            A("  #pragma unroll")
            A("  for (int r = 0; r < 6; r++) { Spi[r] = Sn[r]; }")
            if i > 0:
                A("  #pragma unroll")
                A("  for (int r = 0; r < 6; r++) { Sn[r] = s_Sp[%d + r]; }" % (6 * (i - 1)))
            A("  *(act%d ? &s_Mc[mstart + plen - %d] : s_trash) = %s; // (ancestors in ascending order, then the diagonal)" % (i, i + 1, _MKJ_FAM if bfam else "grid_dot6(Spi, t1)"))
            A("  GRID_SCHED_FENCE(); }")
        self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)
    # ------------------------------------------------------------------ factorisation per component shape
    TS(5)
    shapes = P["shapes"]
    NCmax = max(len(s_) for s_ in shapes)
    ancs = [_anc_local(s_) for s_ in shapes]
    with_rhs = mode in ("fdgrad", "fd")
    park = mode in ("fdgrad", "minv")
    R32 = (lambda e: "static_cast<T>(static_cast<float>(%s))" % e) if "factor" in tuple(self.tuning.get("round_probe", ())) else (lambda e: e)
    by_branch = branch_factor_by_branch(self)
    if by_branch:
        _emit_factor_by_branch(self, P, with_rhs, R32, use_thread_group, ptr)
    else:
        A("// tree-sparse U D U^T of the joint-space inertia of this lane's component (leaves first: no fill-in), wave-uniform inside the component,")
        A("// fused with the forward substitution of tau - c; the first lane of the component parks the factors in the (now free) X(q) storage")
        with_rhs = mode in ("fdgrad", "fd")
        park = mode in ("fdgrad", "minv")
        if with_rhs:
            A("T bq[%d];" % NCmax)
        for si, sig in enumerate(shapes):
            Nc = len(sig)
            an = ancs[si]
            midx = {}
            for k in range(Nc):
                for i in an[k] + [k]:
                    midx[(i, k)] = len(midx)
            A("%sif (shape == %d) { // component of %d joints" % ("" if si == 0 else "else ", si, Nc), True)
            for k in range(Nc):
                A(" ".join("T A%d_%d = s_Mc[ubase + %d];" % (i, k, midx[(i, k)]) for i in an[k] + [k]))
            if with_rhs:
                A("#pragma unroll")
                A("for (int i = 0; i < %d; i++) { bq[i] = s_qdd[cbase + i]; }" % Nc)
            for k in range(Nc - 1, -1, -1):
                R32 = (lambda e: "static_cast<T>(static_cast<float>(%s))" % e) if "factor" in tuple(self.tuning.get("round_probe", ())) else (lambda e: e)
                A("const T rd%d = %s;" % (k, R32("grid_rcp(A%d_%d)" % (k, k))))
                if an[k]:
                    A(" ".join("const T U%d_%d = %s;" % (i, k, R32("A%d_%d*rd%d" % (i, k, k))) for i in an[k]))
                    for j in an[k]:
                        A(" ".join("A%d_%d -= U%d_%d*A%d_%d;" % (i, j, i, k, j, k) for i in an[k] if i <= j))
                    if with_rhs:
                        A(" ".join("bq[%d] -= U%d_%d*bq[%d];" % (i, i, k, k) for i in an[k]))
                if with_rhs:
                    A("bq[%d] *= rd%d;" % (k, k))
            if with_rhs:
                for k in range(1, Nc):
                    if an[k]:
                        A("bq[%d] -= %s;" % (k, " + ".join("U%d_%d*bq[%d]" % (i, k, i) for i in an[k])))
            if park:
                A("if (li == 0) { // the first lane of the component parks the factors", True)
                for k in range(Nc):
                    A(" ".join(["s_Uc[%d] = U%d_%d;" % (midx[(i, k)], i, k) for i in an[k]] + ["s_Uc[%d] = rd%d;" % (midx[(k, k)], k)]))
                self.gen_add_end_control_flow()
            self.gen_add_end_control_flow()
        self.gen_add_sync(use_thread_group)  # every lane has read tau - c
        if with_rhs:
            for si, sig in enumerate(shapes):
                A("%sif (shape == %d && li == 0) {" % ("" if si == 0 else "else ", si), True)
                A("#pragma unroll")
                A("for (int i = 0; i < %d; i++) { s_qdd[cbase + i] = bq[i]; }" % len(sig))
                self.gen_add_end_control_flow()
            self.gen_add_sync(use_thread_group)
    if mode == "fd":
        self.gen_add_end_function()
        return
    if mode == "minv":
        A("// M^-1: the lane of joint j solves M x = e_j with the parked factors and writes row j (= column j)")
        for si, sig in enumerate(shapes):
            Nc = len(sig)
            an = ancs[si]
            midx = {}
            for k in range(Nc):
                for i in an[k] + [k]:
                    midx[(i, k)] = len(midx)
            A("%sif (shape == %d) {" % ("" if si == 0 else "else ", si), True)
            A("T x[%d];" % Nc)
            A("#pragma unroll")
            A("for (int i = 0; i < %d; i++) { x[i] = (i == li) ? static_cast<T>(1) : Z; }" % Nc)
            nf4 = (len(midx) + 3) // 4
            preload = self.tuning["factor_preload"] and 4 * nf4 + Nc <= 200
            if preload:
                A("T F[%d]; // the component's factors, read once with 16-byte loads (its block starts at a multiple of 4 values)" % (4 * nf4))
                A("#pragma unroll")
                A("for (int q = 0; q < %d; q++) { __builtin_memcpy(&F[4*q], __builtin_assume_aligned(s_Uc + 4*q, 16), 4*sizeof(T)); }" % nf4)
            UC = (lambda i_: "F[%d]" % i_) if preload else (lambda i_: "s_Uc[%d]" % i_)
            for k in range(Nc - 1, 0, -1):
                for i in an[k]:
                    A("x[%d] -= %s*x[%d];" % (i, UC(midx[(i, k)]), k))
            for k in range(Nc):
                A("x[%d] *= %s;" % (k, UC(midx[(k, k)])))
            for k in range(1, Nc):
                for i in an[k]:
                    A("x[%d] -= %s*x[%d];" % (k, UC(midx[(i, k)]), i))
            A("#pragma unroll")
            A("for (int i = 0; i < %d; i++) { s_Minv[%d*jid + cbase + i] = x[i]; }" % (Nc, ld))
            self.gen_add_end_control_flow()
        self.gen_add_end_function()
        return

    def column_solves(parked):
        """df/du = -M^-1 dc/du for the two columns this lane owns (the factors are read once for both).  parked (the half-image form): d/dq comes from
        the image, d/dqd from this lane's parked copy; -d/dq goes back to the image, -d/dqd stays in registers (yk) until the first half has left."""
        A("// df/du = -M^-1 dc/du for the two columns this lane owns (rows of its component; all other rows of the column are zero)")
        if parked:
            A("T yk[%d]; // this lane's finished d/dqd column, kept until the d/dq half has left the image" % NCmax)
            A("#pragma unroll")
            A("for (int i = 0; i < %d; i++) { yk[i] = Z; }" % NCmax)
        for si, sig in enumerate(shapes):
            Nc = len(sig)
            an = ancs[si]
            midx = {}
            for k in range(Nc):
                for i in an[k] + [k]:
                    midx[(i, k)] = len(midx)
            A("%sif (shape == %d) {" % ("" if si == 0 else "else ", si), True)
            A("T x[%d], y[%d];" % (Nc, Nc))
            nf4 = (len(midx) + 3) // 4
            preload = self.tuning["factor_preload"] and 4 * nf4 + 2 * Nc <= 200
            if preload:
                A("T F[%d]; // the component's factors, read once with 16-byte loads (its block starts at a multiple of 4 values)" % (4 * nf4))
                A("#pragma unroll")
                A("for (int q = 0; q < %d; q++) { __builtin_memcpy(&F[4*q], __builtin_assume_aligned(s_Uc + 4*q, 16), 4*sizeof(T)); }" % nf4)
            UC = (lambda i_: "F[%d]" % i_) if preload else (lambda i_: "s_Uc[%d]" % i_)
            A("#pragma unroll")
            if parked:
                A("for (int i = 0; i < %d; i++) { x[i] = s_df_du[jid*%d + cbase + i]; y[i] = s_Y[jid*%d + i]; }" % (Nc, n, NCmax))
            else:
                A("for (int i = 0; i < %d; i++) { x[i] = s_df_du[jid*%d + cbase + i]; y[i] = s_df_du[(%d + jid)*%d + cbase + i]; }" % (Nc, n, n, n))
            for k in range(Nc - 1, 0, -1):
                for i in an[k]:
                    A("{ const T uu = %s; x[%d] -= uu*x[%d]; y[%d] -= uu*y[%d]; }" % (UC(midx[(i, k)]), i, k, i, k))
            for k in range(Nc):
                A("{ const T rr = %s; x[%d] *= rr; y[%d] *= rr; }" % (UC(midx[(k, k)]), k, k))
            for k in range(1, Nc):
                for i in an[k]:
                    A("{ const T uu = %s; x[%d] -= uu*x[%d]; y[%d] -= uu*y[%d]; }" % (UC(midx[(i, k)]), k, i, k, i))
            A("#pragma unroll")
            if parked:
                A("for (int i = 0; i < %d; i++) { s_df_du[jid*%d + cbase + i] = -x[i]; yk[i] = -y[i]; }" % (Nc, n))
            else:
                A("for (int i = 0; i < %d; i++) { s_df_du[jid*%d + cbase + i] = -x[i]; s_df_du[(%d + jid)*%d + cbase + i] = -y[i]; }" % (Nc, n, n, n))
            self.gen_add_end_control_flow()

    def store_half(dst_off):
        """half-image form: the image (one half of the record, n^2 values) leaves with 16-byte stores of this lane group."""
        self.gen_add_sync(use_thread_group)
        A("if (d_df_du_k != nullptr) {", True)
        A("for (int e = 4*lane; e + 3 < %d; e += %d) { T tmp[4]; __builtin_memcpy(tmp, __builtin_assume_aligned(s_df_du + e, 4*sizeof(T) < 16 ? 4*sizeof(T) : 16), 4*sizeof(T)); grid_store4(d_df_du_k + %d + e, tmp); }" % (n * n, 4 * lanes, dst_off))
        self.gen_add_end_control_flow()

    if stream and not owner:
        A("// half-image form: pass 1 assembled dc/dqd in the image; every lane parks its own column (rows of its component), pass 2 then overwrites")
        A("// exactly the same entries with dc/dq (same index pattern), so the image needs no clearing in between")
        for si, sig in enumerate(shapes):
            A("%sif (shape == %d) {" % ("" if si == 0 else "else ", si), True)
            A("#pragma unroll")
            A("for (int i = 0; i < %d; i++) { s_Y[jid*%d + i] = s_df_du[jid*%d + cbase + i]; }" % (len(sig), NCmax, n))
            self.gen_add_end_control_flow()
        self.gen_add_sync(use_thread_group)
    # ------------------------------------------------------------------ pass 2
    TS(6)
    A("// the acceleration-dependent parts: a += sum over the ancestors of S_i qdd_i, f^C += sum over the subtree of I_k da_k")
    A("{", True)
    A("T da[6] = {Z, Z, Z, Z, Z, Z}, Ida[6];")
    if owner:
        A("{ const T qddo = active ? s_qdd[js] : Z;")
        A("  #pragma unroll")
        A("  for (int r = 0; r < 6; r++) { da[r] = S[r]*qddo; } }")
        A("grid_prefix_sum(da, mku);")
        hand_down("da", "the qdd part of the acceleration")
    else:
        A("{", True)
        walk_open(True)
        for i in range(D - 1, -1, -1):
            walk_step(i, True)
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { da[r] += Spi[r]*qddi; }")
            A("(void)qdi;")
            walk_close()
        self.gen_add_end_control_flow()
    A("grid_rbi_mul(Ida, I, da); grid_suffix_sum(Ida, mkd);")
    for lv in range(maxlevel, 0, -1):
        A("{ // tree level %d -> %d" % (lv, lv - 1), True)
        A("T y[6]; grid_force_up(y, TR, Tp, Ida);")
        A("if (active && level == %d && pos == 0) {" % lv, True)
        A("#pragma unroll")
        A("for (int r = 0; r < 6; r++) { s_G[28*slot + r] = y[r]; }")
        self.gen_add_end_control_flow()
        self.gen_add_sync(use_thread_group)
        for c in range(maxchild):
            A("{ const int cs = cs%d;" % c)
            A("  if (level == %d && cs >= 0) {" % (lv - 1), True)
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { Ida[r] += s_G[28*cs + r]; }")
            self.gen_add_end_control_flow()
            A("}")
        self.gen_add_end_control_flow()
    A("#pragma unroll")
    A("for (int r = 0; r < 6; r++) { a[r] += da[r]; fC[r] += Ida[r]; }")
    self.gen_add_end_control_flow()
    A("{", True)
    A("T Pdd[6]; grid_mxm(Pdd, a, S); grid_mxm_peq(Pdd, v, Pd);")
    A("T t3[6]; grid_bmul(t3, BC, Pd); grid_rbi_mul_peq(t3, IC, Pdd, static_cast<T>(1)); grid_fxv_peq(t3, S, fC);")
    _probe(self, "t3", "Pdd:6", "t3:6", "a:6", "fC:6")
    if owner:
        A("T t2[6]; grid_bmul(t2, BC, S); grid_rbi_mul_peq(t2, IC, Pd, static_cast<T>(2));")
        _probe(self, "t24", "t2:6", "t4:3")
        if stream:
            A("// (half-image form: the d/dqd entries go straight into the parked columns - column-major, rows of the column's component only)")

        def body(i):
            A("const T up_q = grid_dot6(Spi, t3), lo_q = grid_dot6(t1, Pddi) + t4[0]*Pdi[0] + t4[1]*Pdi[1] + t4[2]*Pdi[2];")
            A("const T up_d = grid_dot6(Spi, t2), lo_d = static_cast<T>(2)*grid_dot6(t1, Pdi) + t4[0]*Spi[0] + t4[1]*Spi[1] + t4[2]*Spi[2];")
            A("*(act%d ? &s_df_du[jid*%d + pj%d] : s_trash) = up_q;" % (i, n, i))
            A("*((act%d && own != %d) ? &s_df_du[pj%d*%d + jid] : s_trash) = lo_q;" % (i, i, i, n))
            if stream:
                A("*(act%d ? &s_Y[jid*%d + pj%d - cbase] : s_trash) = up_d + ((own == %d) ? damping : Z); // + damping on the diagonal (oracle _test.py:486)" % (i, NCmax, i, i))
                A("*((act%d && own != %d) ? &s_Y[pj%d*%d + jid - cbase] : s_trash) = lo_d;" % (i, i, i, NCmax))
            else:
                A("*(act%d ? &s_df_du[(%d + jid)*%d + pj%d] : s_trash) = up_d + ((own == %d) ? damping : Z); // + damping on the diagonal (oracle _test.py:486)" % (i, n, n, i, i))
                A("*((act%d && own != %d) ? &s_df_du[(%d + pj%d)*%d + jid] : s_trash) = lo_d;" % (i, i, n, i, n))

        owner_walk(("S", "Pd", "Pdd"), ("t1", "t2", "t3"), ("t4",), body)
    else:
        A("T vr[6] = {Z, Z, Z, Z, Z, Z}, ar[6] = {Z, Z, Z, gvec[0], gvec[1], gvec[2]};")
        walk_open(True)
        for i in range(D - 1, -1, -1):
            walk_step(i, True)
            A("T Pdi[6], Pddi[6];")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { vr[r] += Spi[r]*qdi; }")
            A("grid_mxm(Pdi, vr, Spi);")
            A("#pragma unroll")
            A("for (int r = 0; r < 6; r++) { ar[r] += Pdi[r]*qdi + Spi[r]*qddi; }")
            A("grid_mxm(Pddi, ar, Spi); grid_mxm_peq(Pddi, vr, Pdi);")
            A("const T up_q = grid_dot6(Spi, t3), lo_q = grid_dot6(t1, Pddi) + t4[0]*Pdi[0] + t4[1]*Pdi[1] + t4[2]*Pdi[2];")
            A("*(act%d ? &s_df_du[jid*%d + pj%d] : s_trash) = up_q;" % (i, n, i))
            A("*((act%d && own != %d) ? &s_df_du[pj%d*%d + jid] : s_trash) = lo_q;" % (i, i, i, n))
            walk_close()
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)
    # ------------------------------------------------------------------ column solves
    TS(7)
    column_solves(stream)
    if stream:
        store_half(0)
        self.gen_add_sync(use_thread_group)  # (the reads of that store are ahead of the writes below in the wave's LDS queue anyway: a compiler fence on the GPU)
        A("// now the d/dqd half: same sparsity, every entry inside the component blocks is overwritten, everything else is still zero")
        for si, sig in enumerate(shapes):
            A("%sif (shape == %d) {" % ("" if si == 0 else "else ", si), True)
            A("#pragma unroll")
            A("for (int i = 0; i < %d; i++) { s_df_du[jid*%d + cbase + i] = yk[i]; }" % (len(sig), n))
            self.gen_add_end_control_flow()
        store_half(n * n)
    if ts_mode:
        self.gen_add_sync(use_thread_group)
        TS(8)
        A("if (lane == 0) { for (int i = 1; i < 9; i++) { s_df_du[i] = static_cast<T>(static_cast<float>(grid_ts[i] - grid_ts[0])); } s_df_du[0] = Z; }")
    self.gen_add_end_function()


def gen_forward_dynamics_gradient_inner_branch_function_call(self, use_thread_group=False, s_df_du_name="s_df_du"):
    self.gen_add_code_line("forward_dynamics_gradient_inner_branch<T>(%s, s_qd, s_u, s_X, s_SP, s_qdd, d_robotModel, gravity, lane);" % s_df_du_name)
