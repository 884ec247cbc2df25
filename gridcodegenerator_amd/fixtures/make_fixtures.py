#!/usr/bin/env python3
"""Writes the robot fixtures (JSON) used by the generator, the oracle and the tests.

No URDF files ship with the reference and URDFParser is not in the build container
(SURVEY.md section 8(c)), so these are hand-authored / synthetic descriptions:
  * iiwa14  - 7-DoF serial chain, all joints revolute-z; values recalled from the public
              iiwa14.urdf (UNVERIFIED - parity is defined against the
              oracle on this same fixture, throughput depends on the topology only)
  * hyq     - 12-DoF quadruped-like tree (4 legs x [HAA about x, HFE about y, KFE about y]), synthetic
  * tree12  - 12-DoF tree with nested branching and a second base-rooted component, synthetic
  * atlas   - 30-DoF humanoid-like tree (back 3, neck 1, arms 2x7, legs 2x6), mixed axes, synthetic
  * mixed5  - 5-DoF branched robot with prismatic joints (edge cases for joint models)
  * arm6    - 6-DoF serial chain, revolute joints about mixed axes, oblique joint frames, full inertia tensors (synthetic)
  * chain12 - 12-DoF serial chain, revolute joints about mixed axes (lane group of 16; synthetic)
  * chain8  - 8-DoF serial chain: exactly as many joints as lanes in the lane group (no spare lane; synthetic)
Run once; the JSON files are committed.
"""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PI = math.pi


def link(name, mass, com, diag, off=(0.0, 0.0, 0.0)):
    return dict(name=name, mass=mass, com=list(com), inertia=[diag[0], off[0], off[1], diag[1], off[2], diag[2]])


def iiwa14():
    origins = [((0, 0, 0.1575), (0, 0, 0)), ((0, 0, 0.2025), (PI / 2, 0, PI)), ((0, 0.2045, 0), (PI / 2, 0, PI)),
               ((0, 0, 0.2155), (PI / 2, 0, 0)), ((0, 0.1845, 0), (-PI / 2, PI, 0)), ((0, 0, 0.2155), (PI / 2, 0, 0)),
               ((0, 0.081, 0), (-PI / 2, PI, 0))]
    masses = [5.76, 6.35, 3.5, 3.5, 3.5, 1.8, 1.2]
    coms = [(0, -0.03, 0.12), (0.0003, 0.059, 0.042), (0, 0.03, 0.13), (0, 0.067, 0.034), (0.0001, 0.021, 0.076),
            (0, 0.0006, 0.0004), (0, 0, 0.02)]
    diags = [(0.033, 0.0333, 0.0123), (0.0305, 0.0304, 0.011), (0.025, 0.0238, 0.0076), (0.017, 0.0164, 0.006),
             (0.01, 0.0087, 0.00449), (0.0049, 0.0047, 0.0036), (0.001, 0.001, 0.001)]
    lims = [2.967, 2.094, 2.967, 2.094, 2.967, 2.094, 3.054]
    joints = []
    for i in range(7):
        joints.append(dict(name="iiwa_joint_%d" % (i + 1), type="revolute", axis="z",
                           parent_link="base" if i == 0 else "iiwa_link_%d" % i,
                           xyz=list(origins[i][0]), rpy=list(origins[i][1]), damping=0.5, limits=[-lims[i], lims[i]],
                           link=link("iiwa_link_%d" % (i + 1), masses[i], coms[i], diags[i])))
    return dict(name="iiwa14", base_link="base", joints=joints)


def hyq():
    joints = []
    for leg, (sx, sy) in zip(["lf", "rf", "lh", "rh"], [(1, 1), (1, -1), (-1, 1), (-1, -1)]):
        joints.append(dict(name=leg + "_haa", type="revolute", axis="x", parent_link="base",
                           xyz=[0.3735 * sx, 0.207 * sy, 0.0], rpy=[0, 0, 0], damping=0.0, limits=[-1.22, 0.44],
                           link=link(leg + "_hipassembly", 2.93, (0.04263 * sx, 0.0 * sy, 0.16931 * 0.1), (0.005495, 0.087136, 0.089871), (0.00007 * sy, -0.0005 * sx, 0.0001))))
        joints.append(dict(name=leg + "_hfe", type="revolute", axis="y", parent_link=leg + "_hipassembly",
                           xyz=[0.08 * sx, 0.0, 0.0], rpy=[0, 0.0, 0], damping=0.0, limits=[-0.87, 1.22],
                           link=link(leg + "_upperleg", 2.638, (0.02 * sx, -0.007 * sy, -0.15074), (0.055, 0.0553, 0.0047), (0.0001 * sx * sy, 0.0025 * sx, -0.0003 * sy))))
        joints.append(dict(name=leg + "_kfe", type="revolute", axis="y", parent_link=leg + "_upperleg",
                           xyz=[0.0, 0.0, -0.35], rpy=[0, 0, 0], damping=0.0, limits=[-2.44, -0.36],
                           link=link(leg + "_lowerleg", 0.881, (0.008 * sx, 0.002 * sy, -0.1254), (0.0164, 0.0165, 0.0004), (0.0, 0.0002 * sx, 0.0))))
    return dict(name="hyq", base_link="base", joints=joints)


def atlas():
    rng = np.random.default_rng(20260313)
    joints = []

    def add(name, axis, parent_link, xyz, rpy, mass, scale):
        com = np.round(rng.uniform(-0.5, 0.5, 3) * scale, 4)
        d = np.round(mass * (scale ** 2) * rng.uniform(0.05, 0.25, 3), 5)
        off = np.round(0.1 * float(d.min()) * rng.uniform(-1, 1, 3), 6)
        damping = [0.0, 0.1, 0.05][len(joints) % 3]
        joints.append(dict(name=name, type="revolute", axis=axis, parent_link=parent_link, xyz=list(xyz), rpy=list(rpy),
                           damping=damping, limits=[-2.5, 2.5],
                           link=link(name + "_link", mass, com.tolist(), d.tolist(), off.tolist())))
        return name + "_link"

    # torso chain: pelvis -> ltorso (back_bkz) -> mtorso (back_bky) -> utorso (back_bkx)
    l = add("back_bkz", "z", "pelvis", (-0.0125, 0, 0), (0, 0, 0), 2.27, 0.15)
    l = add("back_bky", "y", l, (0, 0, 0.162), (0, 0, 0), 0.8, 0.1)
    utorso = add("back_bkx", "x", l, (0, 0, 0.05), (0, 0, 0), 52.0, 0.4)
    # arms (7 each), attached to the upper torso
    for side, s in (("l", 1.0), ("r", -1.0)):
        p = add(side + "_arm_shz", "z", utorso, (0.1406, 0.2256 * s, 0.4776), (0, 0, 0), 4.4, 0.15)
        p = add(side + "_arm_shx", "x", p, (0, 0.11 * s, -0.245), (0, 0, 0), 3.0, 0.15)
        p = add(side + "_arm_ely", "y", p, (0, 0.187 * s, 0.016), (0, 0, 0), 4.5, 0.15)
        p = add(side + "_arm_elx", "x", p, (0, 0.119 * s, 0.0092), (0, 0, 0), 3.4, 0.15)
        p = add(side + "_arm_wry", "y", p, (0, 0.2 * s, -0.0092), (0, 0, 0.3 * s), 2.5, 0.1)
        p = add(side + "_arm_wrx", "x", p, (0, 0.119 * s, 0.0092), (0, 0, 0), 0.8, 0.08)
        p = add(side + "_arm_wry2", "y", p, (0, 0.1 * s, 0), (0.2, 0, 0), 0.6, 0.08)
    # neck
    add("neck_ry", "y", utorso, (0.2546, 0, 0.6215), (0, 0, 0), 1.42, 0.1)
    # legs (6 each), attached to the pelvis
    for side, s in (("l", 1.0), ("r", -1.0)):
        p = add(side + "_leg_hpz", "z", "pelvis", (0, 0.089 * s, 0), (0, 0, 0), 2.4, 0.1)
        p = add(side + "_leg_hpx", "x", p, (0, 0, 0), (0, 0, 0), 1.6, 0.1)
        p = add(side + "_leg_hpy", "y", p, (0.05, 0.0225 * s, -0.066), (0, 0, 0), 8.2, 0.3)
        p = add(side + "_leg_kny", "y", p, (-0.05, 0, -0.374), (0, 0, 0), 4.5, 0.3)
        p = add(side + "_leg_aky", "y", p, (0, 0, -0.422), (0, 0.1, 0), 0.13, 0.05)
        p = add(side + "_leg_akx", "x", p, (0, 0, 0), (0, 0, 0), 2.4, 0.15)
    return dict(name="atlas", base_link="pelvis", joints=joints)


def mixed5():
    return dict(name="mixed5", base_link="base", joints=[
        dict(name="slide_x", type="prismatic", axis="x", parent_link="base", xyz=[0, 0, 0.1], rpy=[0, 0, 0.2], damping=0.3, limits=[-1, 1],
             link=link("cart", 3.0, (0.01, -0.02, 0.03), (0.02, 0.03, 0.025), (0.001, -0.002, 0.0005))),
        dict(name="swing_y", type="revolute", axis="y", parent_link="cart", xyz=[0.05, 0, 0.2], rpy=[0.1, -0.3, 0.4], damping=0.0, limits=[-3, 3],
             link=link("arm_a", 1.5, (0.0, 0.01, 0.25), (0.03, 0.031, 0.002), (0.0, 0.0004, 0.0))),
        dict(name="lift_z", type="prismatic", axis="z", parent_link="arm_a", xyz=[0, 0.02, 0.5], rpy=[0, 0.5, 0], damping=0.1, limits=[-0.5, 0.5],
             link=link("arm_b", 0.9, (0.02, 0.0, 0.1), (0.004, 0.0045, 0.001), (0.0001, 0.0, -0.0002))),
        dict(name="swing_x", type="revolute", axis="x", parent_link="cart", xyz=[-0.05, 0.1, 0.2], rpy=[-0.7, 0.2, 0.0], damping=0.0, limits=[-3, 3],
             link=link("arm_c", 1.1, (0.0, -0.03, 0.2), (0.02, 0.019, 0.003), (0.0, 0.0, 0.0007))),
        dict(name="slide_y", type="prismatic", axis="y", parent_link="arm_c", xyz=[0.0, 0.0, 0.4], rpy=[0.3, 0, -0.6], damping=0.2, limits=[-0.5, 0.5],
             link=link("arm_d", 0.7, (0.01, 0.02, 0.05), (0.002, 0.0025, 0.0011), (0.0, 0.0001, 0.0))),
    ])


def arm6():
    axes = "zyyxyx"
    xyz = [(0, 0, 0.089), (0, 0.136, 0), (0.425, 0, -0.12), (0.392, 0, 0.093), (0, 0.095, 0), (0.03, 0, 0.082)]
    rpy = [(0, 0, 0.3), (0, PI / 2, 0), (0.1, 0, 0), (0, -0.25, 0.4), (PI / 2, 0, 0), (0, 0.2, -0.1)]
    masses = [3.7, 8.393, 2.275, 1.219, 1.219, 0.1879]
    coms = [(0, -0.01, 0.02), (0.2125, 0.01, 0.1360), (0.15, -0.004, 0.0165), (0.002, 0.003, 0.01), (0, 0.01, -0.004), (0.001, -0.002, 0.02)]
    diags = [(0.0102, 0.0103, 0.0067), (0.2269, 0.2270, 0.0151), (0.0494, 0.0495, 0.0041), (0.0021, 0.0022, 0.0021), (0.0021, 0.0021, 0.0022), (0.00013, 0.00014, 0.00019)]
    offs = [(0.0003, -0.0002, 0.0001), (0.004, -0.01, 0.002), (-0.0005, 0.002, 0.0003), (0.0001, 0.0, -0.0001), (0.0, 0.0001, 0.0001), (0.00001, -0.00002, 0.00001)]
    damp = [0.2, 0.0, 0.1, 0.0, 0.05, 0.0]
    joints = []
    for i in range(6):
        joints.append(dict(name="arm6_joint_%d" % (i + 1), type="revolute", axis=axes[i], parent_link="base" if i == 0 else "arm6_link_%d" % i,
                           xyz=list(xyz[i]), rpy=list(rpy[i]), damping=damp[i], limits=[-3.1, 3.1],
                           link=link("arm6_link_%d" % (i + 1), masses[i], coms[i], diags[i], offs[i])))
    return dict(name="arm6", base_link="base", joints=joints)


def chain12(count=12, seed=20261004, name="chain12"):
    rng = np.random.default_rng(seed)
    joints = []
    for i in range(count):
        scale = 0.25 * (0.9 ** i)
        mass = round(6.0 * (0.8 ** i), 4)
        com = np.round(rng.uniform(-0.4, 0.4, 3) * scale, 4)
        d = np.round(mass * scale ** 2 * rng.uniform(0.08, 0.3, 3), 6)
        off = np.round(0.1 * float(d.min()) * rng.uniform(-1, 1, 3), 7)
        xyz = np.round(rng.uniform(-1, 1, 3) * scale, 4)
        rpy = [0.0, 0.0, 0.0] if i % 3 == 0 else np.round(rng.uniform(-0.6, 0.6, 3), 3).tolist()
        joints.append(dict(name="%s_joint_%d" % (name, i + 1), type="revolute", axis="zxy"[i % 3], parent_link="base" if i == 0 else "%s_link_%d" % (name, i),
                           xyz=xyz.tolist(), rpy=rpy, damping=[0.0, 0.1][i % 2], limits=[-3.1, 3.1],
                           link=link("%s_link_%d" % (name, i + 1), mass, com.tolist(), d.tolist(), off.tolist())))
    return dict(name=name, base_link="base", joints=joints)


def chain8():
    return chain12(8, 20261005, "chain8")


def tree12():
    """12-DoF tree with nested branching (trunk 2 -> forearm 2 -> two fingers 2 + 1, and a 3-joint side arm) plus a separate 2-joint tail on the
    base: three tree levels, two base-rooted components of different shape.  Synthetic; exercises the branch-frame path's level loop."""
    rng = np.random.default_rng(20261006)
    joints = []

    def add(name, parent_link, mass, scale):
        axis = "xyz"[int(rng.integers(0, 3))]
        com = np.round(rng.uniform(-0.5, 0.5, 3) * scale, 4)
        d = np.round(mass * (scale ** 2) * rng.uniform(0.05, 0.25, 3), 5)
        off = np.round(0.1 * float(d.min()) * rng.uniform(-1, 1, 3), 6)
        xyz = np.round(rng.uniform(-0.3, 0.3, 3), 3)
        rpy = np.round(rng.uniform(-0.5, 0.5, 3), 3)
        joints.append(dict(name=name, type="revolute", axis=axis, parent_link=parent_link, xyz=xyz.tolist(), rpy=rpy.tolist(),
                           damping=[0.0, 0.2, 0.05][len(joints) % 3], limits=[-3.0, 3.0],
                           link=link(name + "_link", mass, com.tolist(), d.tolist(), off.tolist())))
        return name + "_link"

    t = add("trunk0", "base", 8.0, 0.3)
    t = add("trunk1", t, 5.0, 0.3)
    f = add("fore0", t, 3.0, 0.2)
    f = add("fore1", f, 2.0, 0.2)
    a = add("fingerA0", f, 0.4, 0.08)
    add("fingerA1", a, 0.2, 0.06)
    add("fingerB0", f, 0.3, 0.08)
    s_ = add("side0", t, 2.5, 0.2)
    s_ = add("side1", s_, 1.5, 0.15)
    add("side2", s_, 0.8, 0.1)
    g = add("tail0", "base", 1.2, 0.15)
    add("tail1", g, 0.6, 0.1)
    return dict(name="tree12", base_link="base", joints=joints)


if __name__ == "__main__":
    for fn in (iiwa14, hyq, atlas, mixed5, arm6, chain12, chain8, tree12):
        d = fn()
        with open(os.path.join(HERE, d["name"] + ".json"), "w") as f:
            json.dump(d, f, indent=1)
        print(d["name"], len(d["joints"]), "joints")
