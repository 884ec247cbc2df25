"""Pins the CPU oracle (oracle/rbd_oracle.c) against golden vectors produced by the reference's own
NumPy functions (/root/reference/_test.py, via tests/golden/make_goldens.py)."""
import copy

import numpy as np
import pytest

from gridcodegenerator_amd.robot import RobotModel
from oracle.rbd_oracle import Oracle

CASES = ["iiwa14", "iiwa14_nodamp", "hyq", "atlas", "mixed5", "arm6", "chain12", "chain8", "tree12"]


def robot_for(case):
    r = RobotModel.from_fixture(case.replace("_nodamp", ""))
    if case.endswith("_nodamp"):
        d = copy.deepcopy(r.desc)
        for j in d["joints"]:
            j["damping"] = 0.0
        r = RobotModel(d)
    return r


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("case", CASES)
def test_oracle_f64_matches_reference_goldens(case, golden):
    g = golden(case)
    orc = Oracle(robot_for(case))
    for k in range(g["q"].shape[0]):
        q, qd, u = g["q"][k], g["qd"][k], g["u"][k]
        c, v, a, f = orc.rnea(q, qd)
        assert relerr(c, g["c"][k]) < 1e-12
        assert relerr(v, g["v"][k]) < 1e-12 and relerr(a, g["a"][k]) < 1e-12 and relerr(f, g["f"][k]) < 1e-12
        assert relerr(orc.minv(q, True), g["Minv"][k]) < 1e-10
        assert relerr(orc.minv(q, False), g["Minv_upper"][k]) < 1e-10
        c2, v2, a2, f2 = orc.rnea(q, qd, g["qdd"][k])
        assert relerr(c2, g["c2"][k]) < 1e-12 and relerr(f2, g["f2"][k]) < 1e-12
        assert relerr(orc.rnea_grad(q, qd, g["qdd"][k]), g["dc_du"][k]) < 1e-11
        df, qdd, Minv, dc = orc.fd_grad(q, qd, u, full=True)
        assert relerr(qdd, g["qdd"][k]) < 1e-9
        assert relerr(df, g["df_du"][k]) < 1e-9


@pytest.mark.parametrize("case", ["iiwa14", "hyq", "atlas"])
def test_oracle_f32_within_tolerance(case, golden):
    """The fp32 build is the timed CPU baseline; it must meet the same acceptance band as the GPU path."""
    g = golden(case)
    orc = Oracle(robot_for(case), dtype=np.float32)
    for k in range(g["q"].shape[0]):
        df = orc.fd_grad(g["q"][k], g["qd"][k], g["u"][k])
        assert np.abs(df - g["df_du"][k]).max() <= 1e-4 * np.abs(g["df_du"][k]).max()


def test_batch_layout_matches_device_layout(golden):
    g = golden("iiwa14")
    r = robot_for("iiwa14")
    orc = Oracle(r)
    n = r.n
    x = np.hstack([g["q"], g["qd"], g["u"]])
    out, used = orc.fd_grad_batch(x, nthreads=2)
    assert used == 2
    for k in range(x.shape[0]):
        # d_df_du[k].reshape(2n, n).T == df_du
        assert relerr(out[k].reshape(2 * n, n).T, g["df_du"][k]) < 1e-9


def test_damping_changes_c_and_diag(golden):
    """Damping follows the reference's NumPy oracle (_test.py:103-105,486), pinned by the iiwa14 goldens."""
    g = golden("iiwa14")
    o_d, o_0 = Oracle(robot_for("iiwa14")), Oracle(robot_for("iiwa14_nodamp"))
    q, qd, qdd = g["q"][0], g["qd"][0], g["qdd"][0]
    assert np.allclose(o_d.rnea(q, qd)[0] - o_0.rnea(q, qd)[0], 0.5 * qd)
    d = o_d.rnea_grad(q, qd, qdd) - o_0.rnea_grad(q, qd, qdd)
    assert np.allclose(d[:, :7], 0) and np.allclose(d[:, 7:], 0.5 * np.eye(7))


def test_rnea_grad_matches_finite_differences():
    r = robot_for("hyq")
    orc = Oracle(r)
    rng = np.random.default_rng(5)
    q, qd, qdd = rng.uniform(-1, 1, r.n), rng.uniform(-1, 1, r.n), rng.uniform(-1, 1, r.n)
    dc = orc.rnea_grad(q, qd, qdd)
    eps = 1e-6
    for j in range(r.n):
        e = np.zeros(r.n)
        e[j] = eps
        num_q = (orc.rnea(q + e, qd, qdd)[0] - orc.rnea(q - e, qd, qdd)[0]) / (2 * eps)
        num_qd = (orc.rnea(q, qd + e, qdd)[0] - orc.rnea(q, qd - e, qdd)[0]) / (2 * eps)
        assert np.allclose(dc[:, j], num_q, atol=1e-6)
        assert np.allclose(dc[:, r.n + j], num_qd, atol=1e-6)
