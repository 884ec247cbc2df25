#!/usr/bin/env python3
"""Generates the golden vectors in tests/golden/*.npz by importing the REFERENCE's own NumPy oracle
(/root/reference/_test.py: test_rnea :109, test_minv :213, test_rnea_grad :490, test_fd_grad :496) and
running it on our robot fixtures through the duck-typed RobotModel getters.

Runs only in the build container (the reference does not exist on the GPU box); the .npz outputs are
data (inputs + expected outputs) and are committed.  Usage:  python tests/golden/make_goldens.py [golden name ...]
"""
import contextlib
import copy
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root")  # the reference uses relative imports: import it as package `reference`
sys.dont_write_bytecode = True

from reference.GRiDCodeGenerator import GRiDCodeGenerator as RefGen  # noqa: E402

from gridcodegenerator_amd.robot import RobotModel  # noqa: E402

CASES = [  # (golden name, fixture, number of states, zero the damping?)
    ("iiwa14", "iiwa14", 16, False),
    ("iiwa14_nodamp", "iiwa14", 8, True),
    ("hyq", "hyq", 16, False),
    ("atlas", "atlas", 8, False),
    ("mixed5", "mixed5", 8, False),
    ("arm6", "arm6", 8, False),
    ("chain12", "chain12", 8, False),
    ("chain8", "chain8", 8, False),
    ("tree12", "tree12", 8, False),
]


def sample_inputs(n, count, seed):
    """Same distribution as the benchmark workload."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(-np.pi, np.pi, (count, n))
    qd = rng.uniform(-2.0, 2.0, (count, n))
    u = rng.uniform(-10.0, 10.0, (count, n))
    return q, qd, u


def main():
    only = set(sys.argv[1:])
    for gname, fixture, count, nodamp in CASES:
        if only and gname not in only:
            continue
        robot = RobotModel.from_fixture(fixture)
        if nodamp:
            desc = copy.deepcopy(robot.desc)
            for jd in desc["joints"]:
                jd["damping"] = 0.0
            robot = RobotModel(desc)
        n = robot.get_num_pos()
        ref = RefGen(robot)
        q, qd, u = sample_inputs(n, count, seed=1234 + n)
        out = {k: [] for k in ("c", "v", "a", "f", "Minv", "Minv_upper", "qdd", "c2", "v2", "a2", "f2", "dc_du", "df_du")}
        sink = io.StringIO()  # test_rnea_grad_inner prints unconditionally (_test.py:250-253)
        with contextlib.redirect_stdout(sink):
            for k in range(count):
                c, v, a, f = ref.test_rnea(q[k], qd[k], None)
                Minv = ref.test_minv(q[k], True)
                Minv_upper = ref.test_minv(q[k], False)
                qdd = Minv @ (u[k] - c)
                c2, v2, a2, f2 = ref.test_rnea(q[k], qd[k], qdd)
                dc_du = ref.test_rnea_grad(q[k], qd[k], qdd)
                df_du = ref.test_fd_grad(q[k], qd[k], u[k])
                for key, val in zip(out.keys(), (c, v, a, f, Minv, Minv_upper, qdd, c2, v2, a2, f2, dc_du, df_du)):
                    out[key].append(np.array(val, dtype=np.float64))
        arrays = {k: np.stack(vs) for k, vs in out.items()}
        arrays.update(q=q, qd=qd, u=u, gravity=np.float64(9.81))
        path = os.path.join(HERE, gname + ".npz")
        np.savez_compressed(path, **arrays)
        print("%-14s n=%2d states=%2d  |df_du|max=%.3g |Minv|max=%.3g -> %s" % (
            gname, n, count, np.abs(arrays["df_du"]).max(), np.abs(arrays["Minv"]).max(), os.path.relpath(path, REPO)))


if __name__ == "__main__":
    main()
