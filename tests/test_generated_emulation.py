"""CPU checks of the GENERATED code (no GPU needed): the unchanged generated header and the unchanged C-ABI shim are compiled
with g++ against the thread-per-lane HIP emulation in tests/emu/ and run against the reference-generated goldens.
This validates the generated algorithm, indexing, topology handling and the C-ABI plumbing; the `-m gpu` tests are the parity
tests proper (real wave64 execution, real LDS ordering, real fp32 code generation)."""
import numpy as np
import pytest

from emu_harness import emu_library
from gridcodegenerator_amd import RobotModel

TOL = 1e-4


def per_solve_err(got, ref):
    got = got.reshape(got.shape[0], -1).astype(np.float64)
    ref = ref.reshape(ref.shape[0], -1)
    return (np.abs(got - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-30)).max()


@pytest.fixture(scope="module")
def libs():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = emu_library(name, max_timesteps=128)
        return cache[name]

    return get


@pytest.mark.parametrize("name", ["iiwa14", "hyq", "atlas", "mixed5", "arm6", "chain12", "chain8", "tree12"])
def test_emulated_fd_grad_matches_goldens(name, libs, golden):
    g = golden(name)
    lib = libs(name)
    x = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)
    lib.set_launch_dims(0, 64)
    out = lib.forward_dynamics_gradient_host(x)
    ref = np.stack([g["df_du"][k].T.reshape(-1) for k in range(x.shape[0])])
    assert per_solve_err(out, ref) <= TOL


@pytest.mark.parametrize("name", ["iiwa14", "hyq", "mixed5"])
def test_emulated_one_column_per_lane_variant(name, golden):
    """COLS_PER_LANE=1: d/dq columns on the first half of the lane group, d/dqd columns on the second half, the
    articulated-inertia columns ride on lanes of the second half (all kernels share the lane mapping)."""
    g = golden(name)
    lib = emu_library(name, max_timesteps=64, cols_per_lane=1)
    n = lib.n
    assert lib.lanes_per_solve >= 2 * n
    N = 6
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:N])
    lib.set_launch_dims(0, 64)
    out = lib.forward_dynamics_gradient_host(x)
    assert per_solve_err(out, np.stack([g["df_du"][k].T.reshape(-1) for k in range(N)])) <= TOL
    Minv = np.zeros((N, n * n), np.float32)
    lib.direct_minv_device(x, N, Minv)
    assert per_solve_err(Minv, np.stack([g["Minv_upper"][k].T.reshape(-1) for k in range(N)])) <= TOL
    dc = np.zeros((N, 2 * n * n), np.float32)
    lib.inverse_dynamics_gradient_device(x, np.ascontiguousarray(g["qdd"].astype(np.float32)[:N]), N, dc)
    assert per_solve_err(dc, np.stack([g["dc_du"][k].T.reshape(-1) for k in range(N)])) <= TOL


@pytest.mark.parametrize("name,tuning", [("iiwa14", {"gradient_walk": "lds"}), ("hyq", {"gradient_walk": "lds"}),
                                      ("iiwa14", {"reuse_rnea": True}), ("iiwa14", {"fuse_fd": False}),
                                      ("iiwa14", {"gradient_walk": "registers"}), ("iiwa14", {"tip_chain": "lds"}), ("arm6", {"gradient_walk": "registers"}),
                                      ("iiwa14", {"gradient_walk": "branch"}), ("hyq", {"gradient_walk": "branch"}),
                                      ("atlas", {"gradient_walk": "registers"}), ("tree12", {"stream_out": True}), ("tree12", {"factor_split": "branch"}), ("atlas", {"factor_split": "component"}),
                                      ("hyq", {"gradient_walk": "branch", "factor_split": "branch"}), ("atlas", {"branch_chain": "scan", "branch_walk": "path"}), ("tree12", {"branch_chain": "scan"}),
                                      ("chain12", {"branch_chain": "walk", "branch_walk": "path"}), ("atlas", {"branch_walk": "path"}), ("tree12", {"branch_walk": "owner"}),
                                      ("tree12", {"branch_walk": "owner", "stream_out": True})])
def test_emulated_generation_variants(name, tuning, golden):
    """The non-default generated forms stay correct: LDS-assisted forward accumulation of the derivative walk (what deep trees get),
    RNEA re-use, unfused forward dynamics; the branch-frame path (default for branched revolute robots such as atlas) forced onto chains and
    forests (one branch / equal branches, 8- and 16-lane groups), the column walk forced onto atlas, and the branch-frame kernel that stages one half
    of the record at a time (stream_out: measured slower on the humanoid, kept as an option); the tree-sparse factorisation split by branch (default
    for the humanoid) forced onto the 12-DoF tree (three hand-over levels) and the quadruped, and the per-component form forced onto the humanoid; the scan form of
    the frame chain (default for the 12-joint chain) forced onto the trees and the walk onto the chain; the owner walk (branch_walk: default for the humanoid and the
    12-joint chain) forced onto the 12-DoF tree (three tree levels: two crossings per root path) and the path walk onto the humanoid."""
    g = golden(name)
    lib = emu_library(name, max_timesteps=64, tuning=dict({"so_lanes": "off"}, **tuning))  # (first-order checks only: no second library instance to compile)
    n = lib.n
    N = 5
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:N])
    lib.set_launch_dims(0, 64)
    out = lib.forward_dynamics_gradient_host(x)
    assert per_solve_err(out, np.stack([g["df_du"][k].T.reshape(-1) for k in range(N)])) <= TOL
    dc = np.zeros((N, 2 * n * n), np.float32)
    lib.inverse_dynamics_gradient_device(x, np.ascontiguousarray(g["qdd"].astype(np.float32)[:N]), N, dc)
    assert per_solve_err(dc, np.stack([g["dc_du"][k].T.reshape(-1) for k in range(N)])) <= TOL


@pytest.mark.parametrize("name", ["iiwa14", "arm6", "hyq", "mixed5", "atlas"])
def test_emulated_non_finite_solves_do_not_leak_into_their_neighbours(name, libs, golden):
    """the lane-group scans multiplied the value shifted in from beyond a group's end by a mask of 0 - with 8-lane groups the neighbour in the
    16-lane DPP row is ANOTHER solve, and 0 * NaN (or 0 * Inf) from a diverged trajectory point turned a healthy neighbour's gradient into NaN (the
    reference, one block per solve, keeps solves independent).  8-lane groups now interleave the two solves of a row (GRID_LANE_INTERLEAVE): a row shift
    never reaches the other solve.  A NaN in qd of one solve and an Inf in q of another: every other record stays bit-identical, in every kernel."""
    g = golden(name)
    lib = libs(name)
    n = lib.n
    N = 8
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:N])
    qdd = np.ascontiguousarray(g["qdd"].astype(np.float32)[:N])
    bad = x.copy()
    bad[2, n + 1] = np.nan      # qd of solve 2
    bad[5, 0] = np.inf          # q of solve 5
    bad[7, 2 * n] = 1e38        # u of solve 7: overflows on the way
    keep = [k for k in range(N) if k not in (2, 5, 7)]
    lib.set_launch_dims(0, 64)

    def run(inp):
        outs = [lib.forward_dynamics_gradient_host(inp)]
        for fn, cols, args in ((lib.inverse_dynamics_device, n, (qdd,)), (lib.inverse_dynamics_gradient_device, 2 * n * n, (qdd,))):
            o = np.zeros((N, cols), np.float32)
            fn(inp, *args, N, o)
            outs.append(o)
        for fn, cols in ((lib.forward_dynamics_device, n), (lib.aba_device, n), (lib.direct_minv_device, n * n)):
            o = np.zeros((N, cols), np.float32)
            fn(inp, N, o)
            outs.append(o)
        if lib.has_second_order and n <= 12:
            for fn, args in ((lib.idsva_so_device, (qdd,)), (lib.fdsva_so_device, ())):
                o = np.zeros((N, 4 * n ** 3), np.float32)
                fn(inp, *args, N, o)
                outs.append(o)
        return outs

    clean, dirty = run(x), run(bad)
    lib.set_launch_dims(0, 0)
    for c, d in zip(clean, dirty):
        assert np.isfinite(c).all()
        assert np.array_equal(c[keep], d[keep])
    assert not np.isfinite(dirty[0][2]).all()  # (the poisoned solve itself does come out non-finite)


@pytest.mark.parametrize("blocks,threads", [(1, 64), (2, 32), (1, 16), (3, 48)])
def test_emulated_ragged_launch_dims_and_grid_stride(blocks, threads, libs, golden):
    g = golden("iiwa14")
    lib = libs("iiwa14")
    x = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:13]
    ref = np.stack([g["df_du"][k].T.reshape(-1) for k in range(13)])
    lib.set_launch_dims(blocks, threads)
    out = lib.forward_dynamics_gradient_host(x)
    lib.set_launch_dims(0, 0)
    assert per_solve_err(out, ref) <= TOL


def test_emulated_bad_launch_dims_are_rejected(libs):
    from gridcodegenerator_amd.runtime import GridError

    lib = libs("iiwa14")
    with pytest.raises(GridError):
        lib.set_launch_dims(1, 4)  # fewer threads than one lane group
    with pytest.raises(GridError):
        lib.set_launch_dims(1, 8)  # 8-lane groups interleave pairwise in 16-lane rows (GRID_LANE_INTERLEAVE): the smallest block is one row
    with pytest.raises(GridError):
        lib.set_launch_dims(1, 1024)  # beyond __launch_bounds__
    lib.set_launch_dims(0, 0)


def test_emulated_empty_batch_is_a_noop(libs):
    lib = libs("iiwa14")
    out = lib.forward_dynamics_gradient_host(np.zeros((0, 21), np.float32))
    assert out.shape == (0, 98)


@pytest.mark.parametrize("name", ["iiwa14", "hyq", "atlas", "mixed5", "arm6", "chain12", "chain8", "tree12"])
def test_emulated_component_kernels(name, libs, golden):
    g = golden(name)
    lib = libs(name)
    n = lib.n
    N = g["q"].shape[0]
    lib.set_launch_dims(0, 64)
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32))
    qdd = np.ascontiguousarray(g["qdd"].astype(np.float32))
    c = np.zeros((N, n), np.float32)
    lib.inverse_dynamics_device(x, None, N, c)
    assert per_solve_err(c, g["c"]) <= TOL
    lib.inverse_dynamics_device(x, qdd, N, c)
    assert per_solve_err(c, g["c2"]) <= TOL
    Minv = np.zeros((N, n * n), np.float32)
    lib.direct_minv_device(x, N, Minv)
    assert per_solve_err(Minv, np.stack([g["Minv_upper"][k].T.reshape(-1) for k in range(N)])) <= TOL
    out_qdd = np.zeros((N, n), np.float32)
    lib.forward_dynamics_device(x, N, out_qdd)
    assert per_solve_err(out_qdd, g["qdd"]) <= TOL
    aba_qdd = np.zeros((N, n), np.float32)
    lib.aba_device(x, N, aba_qdd)  # O(n) articulated-body algorithm: same vector as forward dynamics
    assert per_solve_err(aba_qdd, g["qdd"]) <= TOL
    dc = np.zeros((N, 2 * n * n), np.float32)
    lib.inverse_dynamics_gradient_device(x, qdd, N, dc)
    assert per_solve_err(dc, np.stack([g["dc_du"][k].T.reshape(-1) for k in range(N)])) <= TOL
    df = np.zeros((N, 2 * n * n), np.float32)
    lib.forward_dynamics_gradient_qdd_minv_device(x, out_qdd, Minv, N, df)
    assert per_solve_err(df, np.stack([g["df_du"][k].T.reshape(-1) for k in range(N)])) <= TOL
    # compressed input strides (reference USE_COMPRESSED_MEM: stride 2n for q_qd, n for q)
    xq = np.ascontiguousarray(x[:, :2 * n])
    lib.inverse_dynamics_device(xq, qdd, N, c, stride=2 * n)
    assert per_solve_err(c, g["c2"]) <= TOL
    lib.direct_minv_device(np.ascontiguousarray(x[:, :n]), N, Minv, stride=n)
    assert per_solve_err(Minv, np.stack([g["Minv_upper"][k].T.reshape(-1) for k in range(N)])) <= TOL
    lib.set_launch_dims(0, 0)


def test_emulated_single_timing_kernel(libs, golden):
    g = golden("iiwa14")
    lib = libs("iiwa14")
    x = np.hstack([g["q"][5], g["qd"][5], g["u"][5]]).astype(np.float32)  # a state no earlier test left in the device buffer
    lib.forward_dynamics_gradient_host(np.zeros((4, 21), np.float32))
    out, us = lib.forward_dynamics_gradient_single_timing(x, reps=3)
    ref = g["df_du"][5].T.reshape(-1)
    assert np.abs(out - ref).max() <= TOL * np.abs(ref).max()


def test_generator_rejects_unsupported_robots():
    from gridcodegenerator_amd import GRiDCodeGenerator

    class Floating(RobotModel):
        floating_base = True

    with pytest.raises(NotImplementedError):
        GRiDCodeGenerator(Floating(RobotModel.from_fixture("iiwa14").desc))
    with pytest.raises(NotImplementedError):
        GRiDCodeGenerator(RobotModel.from_fixture("iiwa14")).gen_all_code(use_thread_group=True)


def test_emulated_error_behaviour(libs):
    """C-ABI error returns instead of the reference's print-and-exit (GRiDCodeGenerator.py:279-286)."""
    from gridcodegenerator_amd.runtime import GridError

    lib = libs("iiwa14")  # created with max_timesteps=128
    with pytest.raises(GridError):
        lib.forward_dynamics_gradient_host(np.zeros((129, 21), np.float32))  # more solves than grid_init reserved
    with pytest.raises(ValueError):
        lib.forward_dynamics_gradient_host(np.zeros((4, 20), np.float32))  # wrong record length
    with pytest.raises(GridError):
        lib.forward_dynamics_gradient_device(np.zeros((4, 21), np.float32), -1, np.zeros((4, 98), np.float32))  # negative batch
    out = lib.forward_dynamics_gradient_host(np.zeros((1, 21), np.float32))  # still usable afterwards
    assert np.isfinite(out).all()


@pytest.mark.parametrize("name", ["iiwa14", "arm6", "chain8", "hyq", "tree12", "atlas", "mixed5"])
def test_emulated_idsva_so(name, libs, golden):
    """second-order derivatives of inverse dynamics, against the NumPy restatement of the reference's emitter
    (oracle/idsva_so_oracle.py; parity unpinned - the reference holds no vectors for it, see that module).  Serial chains run the tip-frame
    form, the quadruped (a forest), the 12-DoF tree and the 30-DoF humanoid the tree form (reference get_parent_id tables,
    algorithms/_idsva_so.py:171-193,264-284); the humanoid's 432 KB record goes entry by entry to global memory (GRID_SO_DIRECT)."""
    from gridcodegenerator_amd.robot import DuckRobot
    from oracle.idsva_so_oracle import idsva_so

    g = golden(name)
    lib = libs(name)
    n = lib.n
    N = 3
    lib.set_launch_dims(0, 64)
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:N])
    qdd = np.ascontiguousarray(g["qdd"].astype(np.float32)[:N])
    model = DuckRobot(RobotModel.from_fixture(name))
    for use_qdd in (True, False):
        out = np.full((N, 4 * n ** 3), np.nan, np.float32)
        lib.idsva_so_device(x, qdd if use_qdd else None, N, out)
        assert np.isfinite(out).all()  # every entry is written
        for k in range(N):
            ref = idsva_so(model, x[k, :n].astype(np.float64), x[k, n:2 * n].astype(np.float64), qdd[k].astype(np.float64) if use_qdd else np.zeros(n))
            got = out[k].reshape(4, n, n, n)
            for t in range(4):
                assert np.abs(got[t] - ref[t]).max() <= TOL * max(np.abs(ref[t]).max(), 1e-3), (k, t)
    lib.set_launch_dims(0, 0)


@pytest.mark.parametrize("name", ["iiwa14", "tree12"])
def test_emulated_idsva_so_subtree_mapping_variant(name, golden):
    """tuning so_mapping = subtree (lane <-> subtree member, the round-1 mapping) stays available and gives the same tensors as the balanced item
    mapping that ships (round 3: the balanced loops fold the cross products into per-item vectors, so the entries agree to rounding, not bit for bit;
    the round-2 loop bodies - tuning so_loops = mxm - still agree bit for bit)."""
    g = golden(name)
    a = emu_library(name, max_timesteps=8)
    b = emu_library(name, max_timesteps=8, tuning={"so_mapping": "subtree"})
    c = emu_library(name, max_timesteps=8, tuning={"so_loops": "mxm"})
    n, N = a.n, 2
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:N])
    qdd = np.ascontiguousarray(g["qdd"].astype(np.float32)[:N])
    outs = []
    for lib in (a, b, c):
        lib.set_launch_dims(0, 64)
        out = np.full((N, 4 * n ** 3), np.nan, np.float32)
        lib.idsva_so_device(x, qdd, N, out)
        assert np.isfinite(out).all()
        outs.append(out)
    assert np.array_equal(outs[1], outs[2])
    for k in range(N):
        for t in range(4):
            p, r = outs[0][k].reshape(4, -1)[t], outs[1][k].reshape(4, -1)[t]
            assert np.abs(p - r).max() <= 2e-5 * np.abs(r).max()  # (trees: the shipped form also works about every joint's own origin, the variants about the base origin)


def test_emulated_second_order_narrow_and_wide_lane_groups_agree(golden):
    """Robots with 8-lane groups carry a second instance of the library for 16-lane groups (namespace grid::wide, tuning so_lanes) whose kernels the
    second-order entry points launch; tuning so_lanes = off keeps the 8-lane kernels.  Same arithmetic per entry, different lanes execute it."""
    g = golden("iiwa14")
    wide = emu_library("iiwa14", max_timesteps=8)
    narrow = emu_library("iiwa14", max_timesteps=8, tuning={"so_lanes": "off"})
    n, N = wide.n, 3
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:N])
    qdd = np.ascontiguousarray(g["qdd"].astype(np.float32)[:N])
    outs = []
    for lib in (wide, narrow):
        a = np.full((N, 4 * n ** 3), np.nan, np.float32)
        b = np.full((N, 4 * n ** 3), np.nan, np.float32)
        lib.idsva_so_device(x, qdd, N, a)
        lib.fdsva_so_device(x, N, b)
        assert np.isfinite(a).all() and np.isfinite(b).all()
        outs.append((a, b))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.abs(outs[0][1] - outs[1][1]).max() <= 1e-6 * np.abs(outs[1][1]).max()


def test_emulated_second_order_direct_form_on_a_chain(golden):
    """tuning so_direct = True: the second-order kernels write their records straight to global memory and fdsva_so reads the idsva_so tensors back from
    the handle's workspace (what the 30-DoF humanoid runs by necessity), here forced onto the 7-DoF arm's chain form: same values as the LDS-staged form."""
    g = golden("iiwa14")
    staged = emu_library("iiwa14", max_timesteps=8, tuning={"so_lanes": "off"})
    direct = emu_library("iiwa14", max_timesteps=8, tuning={"so_lanes": "off", "so_direct": True})
    n, N = staged.n, 5  # (five solves on 8-lane groups and 64-thread blocks: a ragged last wave)
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:N])
    qdd = np.ascontiguousarray(g["qdd"].astype(np.float32)[:N])
    outs = []
    for lib in (staged, direct):
        a = np.full((N, 4 * n ** 3), np.nan, np.float32)
        b = np.full((N, 4 * n ** 3), np.nan, np.float32)
        lib.idsva_so_device(x, qdd, N, a)
        lib.fdsva_so_device(x, N, b)
        assert np.isfinite(a).all() and np.isfinite(b).all()
        outs.append((a, b))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("name,n", [("iiwa14", 7)])
def test_emulated_second_order_is_refused_where_it_is_not_emitted(name, n):
    """The one-derivative-column-per-lane variant of the library carries no second-order kernels: hipErrorNotSupported.  (Robots with prismatic joints do
    have them since round 3: the tree form carries the joint type through its records - mixed5 in test_emulated_idsva_so / test_emulated_fdsva_so.)"""
    from gridcodegenerator_amd.runtime import GridError

    lib = emu_library(name, max_timesteps=8, cols_per_lane=1)
    assert not lib.has_second_order
    with pytest.raises(GridError):
        lib.idsva_so_device(np.zeros((1, 3 * n), np.float32), None, 1, np.zeros((1, 4), np.float32))
    with pytest.raises(GridError):
        lib.fdsva_so_device(np.zeros((1, 3 * n), np.float32), 1, np.zeros((1, 4), np.float32))


@pytest.mark.parametrize("name", ["iiwa14", "arm6", "hyq", "tree12", "atlas", "chain12", "mixed5"])
def test_emulated_fdsva_so(name, libs, golden):
    """Second half of SURVEY.md section 8(f) rank 3: second-order derivatives of forward dynamics, against the NumPy restatements of the reference's
    idsva_so + fdsva_so emitters fed by the pinned first-order oracle (parity unpinned, see oracle/fdsva_so_oracle.py)."""
    from gridcodegenerator_amd.robot import DuckRobot
    from oracle.fdsva_so_oracle import fdsva_so
    from oracle.idsva_so_oracle import idsva_so
    from oracle.rbd_oracle import Oracle

    g = golden(name)
    lib = libs(name)
    n = lib.n
    N = 3
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:N])
    robot = RobotModel.from_fixture(name)
    model, orc = DuckRobot(robot), Oracle(robot)
    out = np.full((N, 4 * n ** 3), np.nan, np.float32)
    lib.fdsva_so_device(x, N, out)
    assert np.isfinite(out).all()
    for k in range(N):
        q, qd, u = (x[k, i * n:(i + 1) * n].astype(np.float64) for i in range(3))
        df_du, qdd, Minv, _ = orc.fd_grad(q, qd, u, full=True)
        so = np.concatenate([t.reshape(-1) for t in idsva_so(model, q, qd, qdd)])
        ref = fdsva_so(so, Minv, df_du).reshape(4, n, n, n)
        got = out[k].reshape(4, n, n, n)
        for t in range(4):
            assert np.abs(got[t] - ref[t]).max() <= TOL * max(np.abs(ref[t]).max(), 1e-3), (k, t)


@pytest.mark.parametrize("name,tuning", [("atlas", {"so_split": False}), ("hyq", {"so_blocked": False}), ("tree12", {"so_origin": "base"}), ("iiwa14", {"so_hold": 4})])
def test_emulated_fdsva_so_variants_agree_with_the_shipped_form(name, tuning, libs, golden):
    """The forms the shipped second-order kernels replaced stay available as tuning variants and give the same tensors: the single-kernel fdsva_so of the
    30-DoF humanoid (the C ABI runs prepare + contract kernels there; the single fdsva_so_kernel is what a caller launching it himself gets), the dense
    (not block-diagonal) contraction, the tree form about the base origin (fp32: equal to ~1e-4 of the tensor's largest entry only - the reason it was replaced), and the
    compact-record contraction that holds the results of four adjacent k in registers and stores them together (so_hold: less write traffic, more instructions - not shipped)."""
    g = golden(name)
    a = libs(name)
    b = emu_library(name, max_timesteps=8, tuning=tuning)
    n, N = a.n, 2
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:N])
    outs = []
    for lib in (a, b):
        o = np.full((N, 4 * n ** 3), np.nan, np.float32)
        lib.fdsva_so_device(x, N, o)
        assert np.isfinite(o).all()
        outs.append(o.reshape(N, 4, -1))
    tol = 2e-3 if "so_origin" in tuning else 2e-5
    for k in range(N):
        for t in range(4):
            assert np.abs(outs[0][k, t] - outs[1][k, t]).max() <= tol * max(np.abs(outs[0][k, t]).max(), 1e-3)


def _random_tree_description(seed, n):
    """A random fixed-base tree of n revolute joints (ids in DFS pre-order, random axes, offsets, inertias); branches stay below 16 joints."""
    rng = np.random.default_rng(seed)
    kids_left = {}
    joints = []

    def grow(parent_link, budget, run):
        # depth-first: every new joint either continues the current branch, or the parent link gets a further child later
        while budget[0] > 0:
            budget[0] -= 1
            name = "j%d" % len(joints)
            mass = float(rng.uniform(0.2, 5.0))
            scale = float(rng.uniform(0.05, 0.3))
            d = mass * scale ** 2 * rng.uniform(0.05, 0.25, 3)
            off = 0.1 * float(d.min()) * rng.uniform(-1, 1, 3)
            joints.append(dict(name=name, type="revolute", axis="xyz"[int(rng.integers(0, 3))], parent_link=parent_link,
                               xyz=np.round(rng.uniform(-0.3, 0.3, 3), 3).tolist(), rpy=np.round(rng.uniform(-0.6, 0.6, 3), 3).tolist(),
                               damping=float(rng.choice([0.0, 0.1, 0.3])), limits=[-3.0, 3.0],
                               link=dict(name=name + "_link", mass=mass, com=np.round(rng.uniform(-0.5, 0.5, 3) * scale, 4).tolist(),
                                         inertia=[float(d[0]), float(off[0]), float(off[1]), float(d[1]), float(off[2]), float(d[2])])))
            link = name + "_link"
            run += 1
            r = rng.uniform()
            if r < 0.25 and budget[0] > 1:      # branch point: two or three subtrees hang off this link
                for _ in range(int(rng.integers(2, 4))):
                    if budget[0] > 0:
                        grow(link, budget, 0)
                return
            if r < 0.35 or run >= 12:          # leaf
                return
            parent_link = link

    budget = [n]
    while budget[0] > 0:  # several base-rooted components when the first tree ends early
        grow("base", budget, 0)
    return dict(name="rnd%d" % seed, base_link="base", joints=joints)


@pytest.mark.parametrize("seed,n,tuning", [(16, 8, {}), (13, 7, {}), (1, 9, {}), (4, 27, {}),  # (8-lane groups with two tree levels / two components, ..., 32-lane groups)
                                           (58, 20, {"factor_split": "branch", "stream_out": True, "branch_walk": "path"}), (61, 11, {"factor_split": "branch"}),  # (three / four tree levels)
                                           (58, 20, {"branch_walk": "owner", "stream_out": True}), (61, 11, {"branch_walk": "owner", "factor_split": "branch"}),  # (owner walk: up to three crossings per root path)
                                           (4, 27, {"branch_walk": "owner"}), (16, 8, {"branch_walk": "owner"})])  # (auto: owner walk for seeds 1 and 58, path walk for the others)
def test_emulated_random_trees_on_the_branch_frame_path(seed, n, tuning):
    """Generator robustness: random tree topologies (nesting depth, component shapes, lane packing all vary) through the unchanged
    generated header on the branch-frame path, checked against the C oracle (itself pinned by the reference's goldens); the last cases force the
    factorisation split by branch (Schur complements handed up over three and four tree levels) and the half-image form of the kernel; then the owner walk
    (cross-lane reads of the owners' vectors, t-vectors re-expressed at every branch crossing) forced onto deep trees and the path walk onto one that would get the owner walk."""
    from gridcodegenerator_amd import GRiDCodeGenerator
    from oracle.rbd_oracle import Oracle

    robot = RobotModel(_random_tree_description(seed, n))
    gen = GRiDCodeGenerator(robot, tuning=tuning)
    assert gen.branch_frame or gen.tip_frame, "random tree fell back to the column walk"
    lib = emu_library(robot, max_timesteps=16, tuning=dict({"so_lanes": "off"}, **tuning))
    rng = np.random.default_rng(100 + seed)
    N = 5
    x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
    orc = Oracle(robot)
    ref, _ = orc.fd_grad_batch(x.astype(np.float64))
    for blocks, threads in [(0, 64), (1, 128)]:  # one wave per block, and several waves sharing the block's LDS (with a grid-stride loop)
        lib.set_launch_dims(blocks, threads)
        assert per_solve_err(lib.forward_dynamics_gradient_host(x), ref) <= TOL, (blocks, threads)
    lib.set_launch_dims(0, 64)
    if gen.branch_components:
        qdd = rng.uniform(-5, 5, (N, n)).astype(np.float32)
        c = np.zeros((N, n), np.float32)
        lib.inverse_dynamics_device(x, qdd, N, c)
        assert per_solve_err(c, orc.rnea_batch(x[:, :2 * n].astype(np.float64), qdd.astype(np.float64))) <= TOL
        dc = np.zeros((N, 2 * n * n), np.float32)
        lib.inverse_dynamics_gradient_device(x, qdd, N, dc)
        assert per_solve_err(dc, orc.rnea_grad_batch(x[:, :2 * n].astype(np.float64), qdd.astype(np.float64))) <= TOL


def test_emulated_run_longer_than_a_dpp_row_is_split_into_branches():
    """A 20-joint chain does not fit one 16-lane DPP row: the branch-frame plan continues it as a child branch (a junction with one child) and
    the factors no longer fit the X(q) storage (they go behind the path axes).  auto mode leaves such a robot on the column walk (the
    replicated factorisation of a dense 20-joint chain costs more than it saves); tuning={'gradient_walk': 'branch'} forces the path."""
    from gridcodegenerator_amd import GRiDCodeGenerator
    from gridcodegenerator_amd.fixtures.make_fixtures import chain12
    from oracle.rbd_oracle import Oracle

    robot = RobotModel(chain12(20, 77, "chain20"))
    assert not GRiDCodeGenerator(robot).branch_frame
    plan = GRiDCodeGenerator(robot, tuning={"gradient_walk": "branch"}).branch_plan
    assert [len(b) for b in plan["branches"]] == [16, 4] and plan["level"] == [0, 1] and plan["place"]["U"][0] == "sp" and plan["factor_work"] == 1330
    n, N = 20, 3
    rng = np.random.default_rng(0)
    x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
    ref, _ = Oracle(robot).fd_grad_batch(x.astype(np.float64))
    for walk in ("path", "owner"):  # (owner walk: no path records, the factors are all that is left in s_SP)
        lib = emu_library(robot, max_timesteps=8, tuning={"gradient_walk": "branch", "branch_walk": walk})
        lib.set_launch_dims(0, 64)
        assert per_solve_err(lib.forward_dynamics_gradient_host(x), ref) <= TOL, walk


def test_emulated_debug_mode_dumps_the_rnea_intermediates_the_reference_prints(golden, capfd):
    """the reference's comparison mechanism is DEBUG_MODE - the emitted kernels print the same-named intermediates as its NumPy oracle
    (reference algorithms/_inverse_dynamics.py:73-83,137-144,238-252).  The generated RNEA prints s_v / s_a / s_f per joint (forward pass and after the
    backward pass) and c for the first solve of the launch; the numbers are the oracle's v, a, f, c of the goldens."""
    import re

    g = golden("iiwa14")
    lib = emu_library("iiwa14", max_timesteps=4, debug_mode=True, tuning={"so_lanes": "off"})
    n = lib.n
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:1])
    qdd = np.ascontiguousarray(g["qdd"].astype(np.float32)[:1])
    c = np.zeros((1, n), np.float32)
    capfd.readouterr()
    lib.inverse_dynamics_device(x, qdd, 1, c)
    import ctypes
    ctypes.CDLL(None).fflush(None)  # (the C library buffers printf output when stdout is a pipe)
    out = capfd.readouterr().out
    assert np.abs(c[0] - g["c2"][0]).max() <= 1e-4 * np.abs(g["c2"][0]).max()
    vec = lambda tag: [float(t) for t in re.search(re.escape(tag) + r"\n([-0-9. e+]+)\n", out).group(1).split()]
    for j in range(n):
        for nm, tag in (("v2", "s_v[%d]" % j), ("a2", "s_a[%d]" % j), ("f2", "s_f[%d] (after the backward pass)" % j)):
            if nm in g:
                ref = g[nm][0][:, j] if g[nm][0].shape[0] == 6 else g[nm][0][j]
                assert np.abs(np.array(vec(tag)) - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max()), (tag, vec(tag), ref)
        assert ("c[%d] = " % j) in out and ("s_f[%d] (forward pass)" % j) in out
    assert "qdd\n" in out and "X[0] (compact" in out
