"""Numeric robot model exposing the duck-typed getters the generator consumes.

The reference generator is driven by an external ``URDFParser`` robot object
(not vendored with the reference; see SURVEY.md section 8(b).2 for the getter list,
collected from every ``self.robot.<getter>`` call in /root/reference).  This
module provides the same getters from a plain dict / JSON / URDF description so
that (a) our generator, (b) our CPU oracle and (c) the reference's own NumPy
oracle (``/root/reference/_test.py``) can all be driven from one fixture.

Conventions (Featherstone spatial algebra, angular part first):
  * ``X_i(q) = X_J(q_i) * X_tree_i`` maps parent-link motion vectors to link i;
    ``X_tree = [[E, 0], [-E r~, E]]`` with ``E = R(rpy)^T`` and ``r = xyz``.
  * revolute joints: ``X_J = blockdiag(rot_a(q), rot_a(q))`` (rot_a = rx/ry/rz);
    prismatic joints: ``X_J = [[1, 0], [-(q e_a)~, 1]]``.
  * ``S_i`` is a unit 6-vector (index 0..2 revolute x/y/z, 3..5 prismatic x/y/z).
  * spatial inertia about the link frame
    ``I = [[Ic + m c~ c~^T, m c~], [m c~^T, m 1]]``.
  * joints are numbered in DFS pre-order (parent < child, every subtree a
    contiguous id range) - required by the Minv forward sweep
    (reference ``algorithms/_direct_minv.py:182,371``).
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

_AXIS_INDEX = {"x": 0, "y": 1, "z": 2}
FIXTURE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures")


def skew(v: Sequence[float]) -> np.ndarray:
    x, y, z = v
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def rot_axis(axis: int, q: float) -> np.ndarray:
    """Featherstone rx/ry/rz: coordinate transform for a rotation of the frame by q."""
    c, s = math.cos(q), math.sin(q)
    if axis == 0:
        return np.array([[1.0, 0, 0], [0, c, s], [0, -s, c]])
    if axis == 1:
        return np.array([[c, 0, -s], [0, 1.0, 0], [s, 0, c]])
    return np.array([[c, s, 0], [-s, c, 0], [0, 0, 1.0]])


def rpy_to_R(rpy: Sequence[float]) -> np.ndarray:
    """URDF fixed-axis roll/pitch/yaw -> rotation taking child coords to parent coords."""
    r, p, y = rpy
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def plux(E: np.ndarray, r: Sequence[float]) -> np.ndarray:
    X = np.zeros((6, 6))
    X[:3, :3] = E
    X[3:, 3:] = E
    X[3:, :3] = -E @ skew(r)
    return X


def spatial_inertia(mass: float, com: Sequence[float], Ic: np.ndarray) -> np.ndarray:
    C = skew(com)
    I = np.zeros((6, 6))
    I[:3, :3] = Ic + mass * C @ C.T
    I[:3, 3:] = mass * C
    I[3:, :3] = mass * C.T
    I[3:, 3:] = mass * np.eye(3)
    return I


class Joint:
    def __init__(self, jid, name, jtype, axis, parent, xyz, rpy, damping, limits):
        self.jid = jid
        self.name = name
        self.jtype = jtype  # 'revolute' | 'prismatic'
        self.axis = axis  # 0,1,2
        self.parent = parent  # parent joint id, -1 for base
        self.xyz = np.asarray(xyz, dtype=float)
        self.rpy = np.asarray(rpy, dtype=float)
        self.damping = float(damping)
        self.limits = tuple(limits) if limits is not None else (-math.pi, math.pi)
        self.S_index = axis if jtype == "revolute" else 3 + axis
        self.X_tree = plux(rpy_to_R(rpy).T, xyz)

    def get_name(self):
        return self.name

    def get_type(self):
        return self.jtype

    def get_joint_limits(self):
        return list(self.limits)

    def get_damping(self):
        return self.damping

    def X_joint(self, q: float) -> np.ndarray:
        if self.jtype == "revolute":
            E = rot_axis(self.axis, q)
            X = np.zeros((6, 6))
            X[:3, :3] = E
            X[3:, 3:] = E
            return X
        d = np.zeros(3)
        d[self.axis] = q
        return plux(np.eye(3), d)

    def X(self, q: float) -> np.ndarray:
        return self.X_joint(q) @ self.X_tree


class Link:
    def __init__(self, lid, name, mass, com, inertia):
        self.lid = lid
        self.name = name
        self.mass = float(mass)
        self.com = np.asarray(com, dtype=float)
        ixx, ixy, ixz, iyy, iyz, izz = inertia
        self.Ic = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]], dtype=float)
        self.I = spatial_inertia(self.mass, self.com, self.Ic)

    def get_name(self):
        return self.name


class RobotModel:
    """Fixed-base kinematic tree with the getters listed in SURVEY.md section 8(b).2."""

    floating_base = False

    def __init__(self, desc: Dict):
        self.name = desc["name"]
        self.desc = desc
        raw = desc["joints"]
        # ---- renumber in DFS pre-order so that every subtree is a contiguous id range
        children: Dict[str, List[int]] = {}
        for k, jd in enumerate(raw):
            children.setdefault(jd["parent_link"], []).append(k)
        base = desc.get("base_link", "base")
        order: List[int] = []
        parent_of: Dict[int, int] = {}

        def visit(link_name, parent_jid):
            for k in children.get(link_name, []):
                jid = len(order)
                order.append(k)
                parent_of[jid] = parent_jid
                visit(raw[k]["link"]["name"], jid)

        visit(base, -1)
        if len(order) != len(raw):
            raise ValueError("robot description is not a single tree rooted at '%s'" % base)
        self.joints: List[Joint] = []
        self.links: List[Link] = []
        for jid, k in enumerate(order):
            jd = raw[k]
            if jd["type"] not in ("revolute", "prismatic"):
                raise ValueError("unsupported joint type %r (merge fixed joints first)" % jd["type"])
            axis = jd["axis"]
            if isinstance(axis, str):
                axis = _AXIS_INDEX[axis]
            self.joints.append(Joint(jid, jd["name"], jd["type"], axis, parent_of[jid],
                                     jd.get("xyz", (0, 0, 0)), jd.get("rpy", (0, 0, 0)),
                                     jd.get("damping", 0.0), jd.get("limits")))
            ld = jd["link"]
            self.links.append(Link(jid, ld["name"], ld["mass"], ld.get("com", (0, 0, 0)), ld["inertia"]))
        self.n = len(self.joints)
        bi = desc.get("base_inertia")
        self.base_I = np.zeros((6, 6)) if bi is None else spatial_inertia(bi["mass"], bi.get("com", (0, 0, 0)), np.diag(bi["inertia_diag"]))
        # ---- topology tables
        self.parent = [j.parent for j in self.joints]
        self.bfs_level = []
        for j in range(self.n):
            self.bfs_level.append(0 if self.parent[j] == -1 else self.bfs_level[self.parent[j]] + 1)
        self.ancestors = []
        for j in range(self.n):
            a = []
            p = self.parent[j]
            while p != -1:
                a.append(p)
                p = self.parent[p]
            self.ancestors.append(sorted(a))
        self.subtree = [[k for k in range(self.n) if k == j or j in self.ancestors[k]] for j in range(self.n)]
        for j in range(self.n):
            st = self.subtree[j]
            assert st == list(range(j, j + len(st))), "joint numbering is not DFS pre-order"
        self.children = [[k for k in range(self.n) if self.parent[k] == j] for j in range(self.n)]

    # ------------------------------------------------------------------ sizes
    def get_num_pos(self):
        return self.n

    def get_num_vel(self):
        return self.n

    def get_num_joints(self):
        return self.n

    def get_num_fixed_joints(self):
        return 0

    # --------------------------------------------------------------- topology
    def get_parent_id(self, jid):
        return self.parent[jid]

    def get_parent_id_array(self):
        return list(self.parent)

    def get_children_by_id(self, jid):
        return list(self.children[jid])

    def is_serial_chain(self):
        return all(self.parent[j] == j - 1 for j in range(self.n))

    def get_bfs_level_by_id(self, jid):
        return self.bfs_level[jid]

    def get_max_bfs_level(self):
        return max(self.bfs_level)

    def get_ids_by_bfs_level(self, level):
        return [j for j in range(self.n) if self.bfs_level[j] == level]

    def get_max_bfs_width(self):
        return max(len(self.get_ids_by_bfs_level(l)) for l in range(self.get_max_bfs_level() + 1))

    def get_ancestors_by_id(self, jid):
        return list(self.ancestors[jid])  # fresh list: reference callers mutate it (_test.py:355-356)

    def get_subtree_by_id(self, jid):
        return list(self.subtree[jid])

    def get_total_ancestor_count(self):
        return sum(len(a) for a in self.ancestors)

    def get_total_subtree_count(self):
        return sum(len(s) for s in self.subtree)

    def get_max_num_ancestors(self):
        return max(len(a) for a in self.ancestors)

    def get_is_in_subtree_of(self, jid, root):
        return jid in self.subtree[root]

    def get_is_ancestor_of(self, jid, of):
        return jid in self.ancestors[of]

    def get_unique_parent_ids(self, inds):
        return sorted(set(self.parent[i] for i in inds))

    def has_repeated_parents(self, inds):
        ps = [self.parent[i] for i in inds]
        return len(set(ps)) != len(ps)

    def get_leaf_nodes(self):
        return [j for j in range(self.n) if not self.children[j]]

    def get_total_leaf_nodes(self):
        return len(self.get_leaf_nodes())

    def get_jid_ancestor_ids(self, include_joint=False):
        return [self.ancestors[j] + ([j] if include_joint else []) for j in range(self.n)]

    # ------------------------------------------------------------ joints/links
    def get_joint_by_id(self, jid):
        return self.joints[jid]

    def get_link_by_id(self, lid):
        return self.links[lid]

    def get_joints_ordered_by_id(self):
        return list(self.joints)

    def get_joint_by_name(self, name):
        for j in self.joints:
            if j.name == name:
                return j
        return None

    def get_damping_by_id(self, jid):
        return self.joints[jid].damping

    def get_S_by_id(self, jid):
        S = np.zeros(6)
        S[self.joints[jid].S_index] = 1.0
        return S

    def get_S_index_by_id(self, jid):
        return self.joints[jid].S_index

    def get_S_inds(self, n=None):
        return [str(self.joints[j].S_index) for j in range(self.n if n is None else n)]

    def are_Ss_identical(self, inds):
        return len(set(self.joints[i].S_index for i in inds)) <= 1

    # ------------------------------------------------------ transforms/inertias
    def get_Xmat_Func_by_id(self, jid):
        joint = self.joints[jid]
        return lambda q: joint.X(float(q))

    def get_Xmat_Funcs_ordered_by_id(self):
        return [self.get_Xmat_Func_by_id(j) for j in range(self.n)]

    def get_Xmats_ordered_by_id(self):
        """sympy 6x6 matrices in the symbol ``theta`` (what the reference generator stringifies,
        reference helpers/_topology_helpers.py:53-58,259-262).  Needs sympy; only used to drive
        the *reference* generator from our fixtures."""
        import sympy as sp

        theta = sp.Symbol("theta")
        out = []
        for j in self.joints:
            c, s = sp.cos(theta), sp.sin(theta)
            if j.jtype == "revolute":
                E = {0: sp.Matrix([[1, 0, 0], [0, c, s], [0, -s, c]]),
                     1: sp.Matrix([[c, 0, -s], [0, 1, 0], [s, 0, c]]),
                     2: sp.Matrix([[c, s, 0], [-s, c, 0], [0, 0, 1]])}[j.axis]
                XJ = sp.diag(E, E)
            else:
                d = [0, 0, 0]
                d[j.axis] = theta
                dx = sp.Matrix([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
                XJ = sp.Matrix(sp.BlockMatrix([[sp.eye(3), sp.zeros(3)], [-dx, sp.eye(3)]]))
            XT = sp.Matrix(6, 6, lambda r, col: sp.Float(float(j.X_tree[r, col])) if j.X_tree[r, col] != 0 else 0)
            out.append(XJ * XT)
        return out

    def get_Imats_ordered_by_id(self):
        """Index 0 is the base inertia (dropped with [1:] by callers, reference _test.py:17)."""
        return [self.base_I.copy()] + [l.I.copy() for l in self.links]

    def get_Imat_by_id(self, jid):
        return self.links[jid].I.copy()

    def get_Imats_dict_by_id(self):
        return {j: self.links[j].I.copy() for j in range(self.n)}

    # ------------------------------------------------------------ constructors
    @staticmethod
    def from_json(path: str) -> "RobotModel":
        with open(path) as f:
            return RobotModel(json.load(f))

    @staticmethod
    def from_fixture(name: str) -> "RobotModel":
        return RobotModel.from_json(os.path.join(FIXTURE_DIR, name + ".json"))

    def to_arrays(self, dtype=np.float64) -> Dict[str, np.ndarray]:
        """Flat tables consumed by the C oracle (oracle/rbd_oracle.c)."""
        n = self.n
        return {
            "n": n,
            "parent": np.asarray(self.parent, dtype=np.int32),
            "S_index": np.asarray([j.S_index for j in self.joints], dtype=np.int32),
            "X_tree": np.ascontiguousarray(np.stack([j.X_tree for j in self.joints]).astype(dtype)),  # [n][row][col]
            "I": np.ascontiguousarray(np.stack([l.I for l in self.links]).astype(dtype)),
            "damping": np.asarray([j.damping for j in self.joints], dtype=dtype),
        }


class DuckRobot:
    """Normalises ANY object that offers the URDFParser-style getters (including RobotModel) into the
    numeric tables the generator needs, without sympy.  ``X_tree_i`` is recovered as ``X_i(0)`` and the
    joint model is cross-checked against the object's own ``get_Xmat_Func_by_id`` at a test angle, so a
    robot our joint models cannot represent fails loudly at generation time."""

    def __init__(self, robot):
        self.robot = robot
        self.name = getattr(robot, "name", "robot")
        if getattr(robot, "floating_base", False):
            raise NotImplementedError("floating-base robots are outside the hot-path scope (SURVEY.md section 8(f))")
        n = self.n = robot.get_num_pos()
        if robot.get_num_vel() != n or robot.get_num_joints() != n:
            raise ValueError("fixed-base robots must have num_pos == num_vel == num_joints")
        self.parent = [int(robot.get_parent_id(j)) for j in range(n)]
        self.S_index = []
        for j in range(n):
            S = np.asarray(robot.get_S_by_id(j), dtype=float).ravel()
            nz = np.nonzero(S)[0]
            if len(nz) != 1 or S[nz[0]] != 1.0:
                raise ValueError("joint %d: S must be a unit vector with a single +1 (got %s)" % (j, S))
            self.S_index.append(int(nz[0]))
        self.X_tree = []
        for j in range(n):
            f = robot.get_Xmat_Func_by_id(j)
            X0 = np.asarray(f(0.0), dtype=float).reshape(6, 6)
            self.X_tree.append(X0)
            for qt in (0.7, -1.3):
                if not np.allclose(self.X(j, qt), np.asarray(f(qt), dtype=float).reshape(6, 6), atol=1e-9):
                    raise ValueError("joint %d: X(q) is not X_J(q)*X(0) for a principal-axis joint" % j)
        # structural zeros of the tree transforms (|x| < 1e-12, e.g. cos(pi/2) = 6e-17 from URDF rpy's) are made exact so that the
        # generator can emit sparsity-specialised transform products; the nonzero pattern of X(q) = X_J(q) X_tree follows from them
        self.X_tree = [np.where(np.abs(X) < 1e-12, 0.0, X) for X in self.X_tree]
        self.X_pattern = [self._x_pattern(j) for j in range(n)]
        Imats = robot.get_Imats_ordered_by_id()
        self.I = [np.asarray(Imats[j + 1], dtype=float).reshape(6, 6) for j in range(n)]
        self.damping = [float(robot.get_damping_by_id(j)) if hasattr(robot, "get_damping_by_id") else 0.0 for j in range(n)]
        # derived topology (DFS pre-order is required)
        self.children = [[k for k in range(n) if self.parent[k] == j] for j in range(n)]
        self.ancestors = []
        for j in range(n):
            if not (-1 <= self.parent[j] < j):
                raise ValueError("joint ids must be ordered parent < child")
            a, p = [], self.parent[j]
            while p != -1:
                a.append(p)
                p = self.parent[p]
            self.ancestors.append(sorted(a))
        self.subtree = [[k for k in range(n) if k == j or j in self.ancestors[k]] for j in range(n)]
        for j in range(n):
            if self.subtree[j] != list(range(j, j + len(self.subtree[j]))):
                raise ValueError("joint ids must be in DFS pre-order (contiguous subtrees)")
        self.roots = [j for j in range(n) if self.parent[j] == -1]
        self.depth = [len(a) for a in self.ancestors]

    def _x_pattern(self, j):
        """(E_nz, B_nz): 3x3 boolean masks of the entries of X_j(q) = [[E,0],[B,E]] that can be non-zero for some q."""
        XT = self.X_tree[j]
        E, B = XT[:3, :3], XT[3:, :3]
        s = self.S_index[j]
        a = s % 3
        r1, r2 = (a + 1) % 3, (a + 2) % 3
        PE, PB = np.zeros((3, 3), bool), np.zeros((3, 3), bool)
        for c in range(3):
            if s < 3:  # revolute: rows r1, r2 of E and of B are mixed by cos/sin, row a is constant
                PE[r1, c] = PE[r2, c] = (E[r1, c] != 0) or (E[r2, c] != 0)
                PE[a, c] = E[a, c] != 0
                PB[r1, c] = PB[r2, c] = (B[r1, c] != 0) or (B[r2, c] != 0)
                PB[a, c] = B[a, c] != 0
            else:      # prismatic: E is constant, rows r1, r2 of B pick up q * rows r2, r1 of E
                PE[:, c] = E[:, c] != 0
                PB[r1, c] = (B[r1, c] != 0) or (E[r2, c] != 0)
                PB[r2, c] = (B[r2, c] != 0) or (E[r1, c] != 0)
                PB[a, c] = B[a, c] != 0
        return PE, PB

    def X_joint(self, j, q):
        si = self.S_index[j]
        if si < 3:
            E = rot_axis(si, q)
            X = np.zeros((6, 6))
            X[:3, :3] = E
            X[3:, 3:] = E
            return X
        d = np.zeros(3)
        d[si - 3] = q
        return plux(np.eye(3), d)

    def X(self, j, q):
        return self.X_joint(j, q) @ self.X_tree[j]

    def is_serial_chain(self):
        return all(self.parent[j] == j - 1 for j in range(self.n))


# --------------------------------------------------------------------------- URDF front end
def load_urdf(path_or_text: str, name: Optional[str] = None) -> RobotModel:
    """Minimal URDF reader: revolute/continuous/prismatic joints about +x/+y/+z, fixed joints merged
    into their parent link (leaf fixed links fold their inertia into the parent)."""
    import xml.etree.ElementTree as ET

    text = path_or_text
    if os.path.exists(path_or_text):
        with open(path_or_text) as f:
            text = f.read()
    root = ET.fromstring(text)

    def vec(s, default):
        return [float(x) for x in s.split()] if s else list(default)

    links = {}
    for l in root.findall("link"):
        inertial = l.find("inertial")
        if inertial is None:
            links[l.get("name")] = dict(mass=0.0, com=[0, 0, 0], Ic=np.zeros((3, 3)))
            continue
        o = inertial.find("origin")
        com = vec(o.get("xyz") if o is not None else None, (0, 0, 0))
        rpy = vec(o.get("rpy") if o is not None else None, (0, 0, 0))
        m = float(inertial.find("mass").get("value"))
        it = inertial.find("inertia")
        g = lambda k: float(it.get(k, 0.0))
        Ic = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        R = rpy_to_R(rpy)
        links[l.get("name")] = dict(mass=m, com=com, Ic=R @ Ic @ R.T)
    joints = []
    child_links = set()
    for j in root.findall("joint"):
        o = j.find("origin")
        a = j.find("axis")
        dyn = j.find("dynamics")
        lim = j.find("limit")
        joints.append(dict(name=j.get("name"), type=j.get("type"), parent_link=j.find("parent").get("link"),
                           child=j.find("child").get("link"),
                           xyz=vec(o.get("xyz") if o is not None else None, (0, 0, 0)),
                           rpy=vec(o.get("rpy") if o is not None else None, (0, 0, 0)),
                           axis=vec(a.get("xyz") if a is not None else None, (1, 0, 0)),
                           damping=float(dyn.get("damping", 0.0)) if dyn is not None else 0.0,
                           limits=(float(lim.get("lower", -math.pi)), float(lim.get("upper", math.pi))) if lim is not None else None))
        child_links.add(joints[-1]["child"])
    base = [n_ for n_ in links if n_ not in child_links]
    if len(base) != 1:
        raise ValueError("URDF must have exactly one root link")
    base = base[0]
    # spatial inertia of every link in its own frame; fixed joints fold child into parent
    I6 = {k: spatial_inertia(v["mass"], v["com"], v["Ic"]) for k, v in links.items()}
    # transform from a link's frame to the frame of the moving link it is rigidly attached to
    attach = {base: (base, np.eye(6))}
    out_joints = []
    by_parent = {}
    for jd in joints:
        by_parent.setdefault(jd["parent_link"], []).append(jd)

    def walk(link):
        owner, X_owner_from_link = attach[link]  # motion transform link-frame <- owner-frame is inverse; keep owner<-link as plux
        for jd in by_parent.get(link, []):
            Xt = plux(rpy_to_R(jd["rpy"]).T, jd["xyz"])  # child <- parent-link coords
            if jd["type"] == "fixed":
                X_child_from_owner = Xt @ X_owner_from_link
                attach[jd["child"]] = (owner, X_child_from_owner)
                I6[owner] = I6[owner] + X_child_from_owner.T @ I6[jd["child"]] @ X_child_from_owner
            else:
                ax = np.asarray(jd["axis"], dtype=float)
                k = int(np.argmax(np.abs(ax)))
                if not np.allclose(ax, np.eye(3)[k]):
                    raise ValueError("joint %s: only +x/+y/+z axes are supported (got %s)" % (jd["name"], ax))
                out_joints.append(dict(jd, X_tree=Xt @ X_owner_from_link, axis=k,
                                       type="revolute" if jd["type"] in ("revolute", "continuous") else jd["type"],
                                       parent_link=owner))
                attach[jd["child"]] = (jd["child"], np.eye(6))
            walk(jd["child"])

    walk(base)
    desc_joints = []
    for jd in out_joints:
        X = jd["X_tree"]
        E = X[:3, :3]
        r = -E.T @ X[3:, :3]  # -E r~ = BL  =>  r~ = -E^T BL
        xyz = [r[2, 1], r[0, 2], r[1, 0]]
        R = E.T
        rpy = [math.atan2(R[2, 1], R[2, 2]), math.atan2(-R[2, 0], math.hypot(R[2, 1], R[2, 2])), math.atan2(R[1, 0], R[0, 0])]
        I = I6[jd["child"]]
        m = I[3, 3]
        com = [I[2, 4] / m, I[0, 5] / m, I[1, 3] / m] if m > 0 else [0, 0, 0]
        C = skew(com)
        Ic = I[:3, :3] - m * C @ C.T
        desc_joints.append(dict(name=jd["name"], type=jd["type"], axis=jd["axis"], parent_link=jd["parent_link"], xyz=xyz, rpy=rpy,
                                damping=jd["damping"], limits=jd["limits"],
                                link=dict(name=jd["child"], mass=m, com=com,
                                          inertia=[Ic[0, 0], Ic[0, 1], Ic[0, 2], Ic[1, 1], Ic[1, 2], Ic[2, 2]])))
    return RobotModel(dict(name=name or root.get("name", "robot"), base_link=base, joints=desc_joints))
