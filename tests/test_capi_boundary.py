"""Boundary behaviour of the C ABI and of the generator's constructor, on the CPU emulation (no GPU needed): error returns instead of
exit(), host-buffer entry points of every algorithm, the single-process multi-device driver, the T = double entry points, and the rule
that nothing in the process environment changes the generated code."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from emu_harness import emu_library
from gridcodegenerator_amd import GRiDCodeGenerator, RobotModel
from gridcodegenerator_amd.runtime import GridError, GridLibrary, MultiGpuGrid

TOL = 1e-4
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_solve_err(got, ref):
    got = got.reshape(got.shape[0], -1).astype(np.float64)
    ref = ref.reshape(ref.shape[0], -1)
    return (np.abs(got - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-30)).max()


def test_grid_init_returns_an_error_code_instead_of_exiting():
    """include/grid_capi.h: 'every function returns 0 or a hipError_t' - an allocation that cannot succeed must come back as a non-zero
    return value with a message; the reference's init_gridData would print GPUassert and exit() the host process
    (reference GRiDCodeGenerator.py:279-286)."""
    lib = emu_library("iiwa14", max_timesteps=8)
    with pytest.raises(GridError) as e:
        GridLibrary(lib.path, device=0, max_timesteps=2 ** 31 - 1)  # ~ 6 TB of buffers
    assert "max_timesteps" in str(e.value) or "memory" in str(e.value).lower()
    with pytest.raises(GridError):
        GridLibrary(lib.path, device=99, max_timesteps=8)  # no such device
    # the interpreter is alive and the library still works
    out = lib.forward_dynamics_gradient_host(np.zeros((2, 21), np.float32))
    assert out.shape == (2, 98)


def test_error_hook_of_the_generated_header_defaults_to_the_references_behaviour(tmp_path):
    """Without GRID_ON_GPU_ERROR the generated gpuAssert prints 'GPUassert: ...' and exits with the error code (what downstream C++ expects)."""
    from gridcodegenerator_amd.runtime import generate_header

    gen = tmp_path / "gen"
    generate_header(RobotModel.from_fixture("iiwa14"), str(gen))
    src = tmp_path / "t.cpp"
    src.write_text('#include "grid.cuh"\nint main() { void *p = nullptr; gpuErrchk(hipMalloc(&p, (size_t)1 << 62)); return 0; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++20", "-O0", "-pthread", "-I" + os.path.join(REPO, "tests", "emu"), "-I" + str(gen), "-x", "c++", str(src), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 2 and "GPUassert" in r.stderr  # hipErrorOutOfMemory = 2


@pytest.mark.parametrize("name", ["iiwa14", "hyq"])
def test_host_entry_points_of_every_algorithm(name, golden):
    """The *_host C-ABI entry points mirror the reference's host wrappers (H2D, launch, D2H, synchronous; compressed strides): same
    results as the goldens the reference's own oracle produced."""
    g = golden(name)
    lib = emu_library(name, max_timesteps=32)
    lib.set_launch_dims(0, 64)
    n = lib.n
    N = 6
    q, qd, u, qdd = (np.ascontiguousarray(g[k][:N].astype(np.float32)) for k in ("q", "qd", "u", "qdd"))
    x3, x2 = np.hstack([q, qd, u]), np.hstack([q, qd])
    col = lambda key: np.stack([g[key][k].T.reshape(-1) for k in range(N)])
    assert per_solve_err(lib.inverse_dynamics_host(x3), g["c"][:N]) <= TOL            # qdd = 0, stride 3n
    assert per_solve_err(lib.inverse_dynamics_host(x2, qdd), g["c2"][:N]) <= TOL      # USE_QDD_FLAG, USE_COMPRESSED_MEM (stride 2n)
    assert per_solve_err(lib.inverse_dynamics_gradient_host(x2, qdd), col("dc_du")) <= TOL
    assert per_solve_err(lib.direct_minv_host(q), col("Minv_upper")) <= TOL           # compressed: stride n
    assert per_solve_err(lib.direct_minv_host(x3), col("Minv_upper")) <= TOL
    out_qdd = lib.forward_dynamics_host(x3)
    assert per_solve_err(out_qdd, g["qdd"][:N]) <= TOL
    assert per_solve_err(lib.forward_dynamics_host(x3, aba=True), g["qdd"][:N]) <= TOL
    Minv = lib.direct_minv_host(q)
    assert per_solve_err(lib.forward_dynamics_gradient_qdd_minv_host(x2, out_qdd, Minv), col("df_du")) <= TOL
    assert per_solve_err(lib.forward_dynamics_gradient_host(x3), col("df_du")) <= TOL
    with pytest.raises(GridError):
        lib.forward_dynamics_host(np.zeros((33, 3 * n), np.float32))  # more than grid_init's max_timesteps
    if lib.has_second_order:
        from gridcodegenerator_amd.robot import DuckRobot
        from oracle.idsva_so_oracle import idsva_so

        so = lib.idsva_so_host(x3[:2], qdd[:2])
        ref = np.concatenate([t.reshape(-1) for t in idsva_so(DuckRobot(RobotModel.from_fixture(name)), q[0].astype(np.float64), qd[0].astype(np.float64), qdd[0].astype(np.float64))])
        assert np.abs(so[0] - ref).max() <= TOL * np.abs(ref).max()
        assert np.isfinite(lib.fdsva_so_host(x3[:2])).all()


@pytest.mark.parametrize("name", ["iiwa14", "tree12"])
def test_double_precision_entry_points_match_the_fp64_goldens(name, golden):
    """Every algorithm through its *_f64 host entry point against the goldens of the reference's own oracle, to rounding level: pins the generated
    algorithms (tip-frame path for the arm, branch-frame path for the 12-DoF tree) independently of fp32 effects."""
    g = golden(name)
    lib = emu_library(name, max_timesteps=16)
    lib.set_launch_dims(0, 64)
    n, N = lib.n, 4
    x = np.hstack([g["q"], g["qd"], g["u"]])[:N].astype(np.float64)
    col = lambda key: np.stack([g[key][k].T.reshape(-1) for k in range(N)])
    out = lib.forward_dynamics_gradient_host_f64(x)
    assert out.dtype == np.float64 and per_solve_err(out, col("df_du")) <= 1e-9
    assert per_solve_err(lib.host_f64("inverse_dynamics", x[:, :2 * n], g["qdd"][:N]), g["c2"][:N]) <= 1e-9
    assert per_solve_err(lib.host_f64("inverse_dynamics_gradient", x[:, :2 * n], g["qdd"][:N]), col("dc_du")) <= 1e-9
    assert per_solve_err(lib.host_f64("direct_minv", x[:, :n]), col("Minv_upper")) <= 1e-9
    assert per_solve_err(lib.host_f64("forward_dynamics", x), g["qdd"][:N]) <= 1e-9
    assert per_solve_err(lib.host_f64("aba", x), g["qdd"][:N]) <= 1e-9


@pytest.mark.parametrize("G,N", [(2, 16), (3, 13), (4, 3)])
def test_single_process_multi_device_driver_splits_one_batch_contiguously(G, N, golden):
    """one process, G handles (one per device), the batch cut into G contiguous ranges of ceil(N/G): the result is
    bit-identical to the one-handle result, also when the split is ragged or a device gets nothing."""
    g = golden("iiwa14")
    single = emu_library("iiwa14", max_timesteps=16)
    single.set_launch_dims(0, 64)
    x = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:N]
    ref = single.forward_dynamics_gradient_host(x)
    multi = MultiGpuGrid(single.path, devices=list(range(G)), max_timesteps=16)  # (the emulation pretends to have 8 devices)
    try:
        for p in multi.parts:
            p.set_launch_dims(0, 64)
        assert [p.lib.grid_device(p.handle) for p in multi.parts] == list(range(G))
        rg = MultiGpuGrid.ranges(N, G)
        assert rg[0][0] == 0 and rg[-1][1] == N and all(rg[i][1] == rg[i + 1][0] for i in range(G - 1))
        out = multi.forward_dynamics_gradient_host(x)
        assert np.array_equal(out, ref)
    finally:
        multi.close()


def test_entry_points_restore_the_callers_device():
    lib = emu_library("iiwa14", max_timesteps=8)
    other = GridLibrary(lib.path, device=3, max_timesteps=8)
    try:
        other.set_launch_dims(0, 64)
        other.forward_dynamics_gradient_host(np.zeros((1, 21), np.float32))
        # the emulation tracks the current device like hipSetDevice/hipGetDevice do: the call above ran "on device 3" and put device 0 back
        assert other.lib.grid_device(other.handle) == 3
        assert other.lib.hipemu_get_device() == 0
    finally:
        other.close()


@pytest.mark.parametrize("N,threads", [(101, 256), (5, 256), (3, 128), (9, 72)])
def test_waves_past_the_end_of_a_ragged_batch_write_nothing(N, threads, golden):
    """in the last block of a ragged batch a wave whose lane groups are all past the batch end computed a negative record count and
    re-stored the tail of the last valid record from ANOTHER wave's LDS staging area (an unsynchronised cross-wave read; on the hardware the
    stale value can land after the owner's store).  Deterministic check on the emulation: mode 2 runs every wave of a block EXCEPT the first,
    so every record the first wave owns must keep the sentinel the output was pre-filled with; mode 1 (waves one after another, last first,
    each with its own barrier) must give the full, correct result."""
    g = golden("iiwa14")
    lib = emu_library("iiwa14", max_timesteps=128)
    reps = -(-N // 16)
    x = np.ascontiguousarray(np.tile(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32), (reps, 1))[:N])
    ref = np.tile(np.stack([g["df_du"][k].T.reshape(-1) for k in range(16)]), (reps, 1))[:N]
    gpb = min(threads // 16 * 2, 32)  # (8-lane groups interleave pairwise in 16-lane rows: blocks are whole rows)
    try:
        for blocks in (0, 1):  # 1: grid-stride, the staging area still holds the previous trip's records
            lib.set_launch_dims(blocks, threads)
            lib.lib.hipemu_set_mode(1)
            out = np.full((N, 98), 7.0, np.float32)
            lib.forward_dynamics_gradient_device(x, N, out)
            assert np.isfinite(out).all() and per_solve_err(out, ref) <= TOL
            lib.lib.hipemu_set_mode(2)
            out = np.full((N, 98), 7.0, np.float32)
            lib.forward_dynamics_gradient_device(x, N, out)
            first_wave = np.array([(k % gpb) < 8 for k in range(N)])  # solves whose lane group sits in the first wave of its block
            assert (out[first_wave] == 7.0).all(), "a wave wrote into records it does not own"
            assert per_solve_err(out[~first_wave], ref[~first_wave]) <= TOL if (~first_wave).any() else True
    finally:
        lib.lib.hipemu_set_mode(0)
        lib.set_launch_dims(0, 0)


def test_environment_variables_do_not_change_the_generated_code(tmp_path):
    """ablation switches used to be read from the environment inside the generator (GRID_DEBUG_STOP=3 emitted a truncated,
    WRONG kernel).  They are constructor arguments now; a stray variable in a user's shell must not matter."""
    code = ("import sys; sys.path.insert(0, %r); from gridcodegenerator_amd import GRiDCodeGenerator, RobotModel; import hashlib, os; os.chdir(%r); "
            "print(hashlib.sha256(GRiDCodeGenerator(RobotModel.from_fixture('iiwa14')).gen_all_code().encode()).hexdigest())" % (REPO, str(tmp_path)))
    clean = {k: v for k, v in os.environ.items() if not k.startswith("GRID_")}
    a = subprocess.check_output([sys.executable, "-c", code], env=clean, text=True).strip()
    dirty = dict(clean, GRID_DEBUG_STOP="3", GRID_NO_WAVE_BARRIER="1", GRID_NO_PINS="1", GRID_GRADIENT_WALK="lds", GRID_COLS_PER_LANE="1", GRID_DPP_ASM="0",
                 GRID_TIP_CHAIN="lds", GRID_FUSE_FD="0", GRID_REUSE_RNEA="1", GRID_MIN_WAVES="4", GRID_SO_UNROLL="1", GRID_NT_STORE="0")
    b = subprocess.check_output([sys.executable, "-c", code], env=dirty, text=True).strip()
    assert a == b


def test_tuning_is_an_explicit_constructor_argument():
    robot = RobotModel.from_fixture("iiwa14")
    assert GRiDCodeGenerator(robot).tip_frame
    assert not GRiDCodeGenerator(robot, tuning={"gradient_walk": "registers"}).tip_frame
    with pytest.raises(ValueError):
        GRiDCodeGenerator(robot, tuning={"no_such_knob": 1})
    with pytest.raises(ValueError):
        GRiDCodeGenerator(robot, tuning={"gradient_walk": "fastest"})
    with pytest.raises(ValueError):  # result-breaking ablation switches need an explicit acknowledgement
        GRiDCodeGenerator(robot, tuning={"debug_stop": 5})
    assert GRiDCodeGenerator(robot, tuning={"debug_stop": 5, "allow_wrong_results": True}).tuning["debug_stop"] == 5


def test_static_shared_memory_mode_is_rejected_not_ignored():
    """reference GRiDCodeGenerator.py:54,61: USE_DYNAMIC_SHARED_MEM=False declares static __shared__ arrays per function; the lane-group kernels
    cannot honour it (their LDS is sized by the launch), so the constructor refuses instead of silently ignoring the flag."""
    with pytest.raises(NotImplementedError):
        GRiDCodeGenerator(RobotModel.from_fixture("iiwa14"), USE_DYNAMIC_SHARED_MEM=False)


def test_device_entry_points_reject_null_pointers_and_short_strides(golden):
    """the *_device entry points validated only the handle, N and the thread count - a NULL buffer or a stride smaller than what the kernel
    loads per solve became a GPU memory fault instead of hipErrorInvalidValue.  (Emulation: the same C-ABI shim, so the checks run without a GPU.)"""
    import ctypes

    from gridcodegenerator_amd.runtime import GridError

    lib = emu_library("iiwa14", max_timesteps=16)
    n = lib.n
    x = np.zeros((4, 3 * n), np.float32)
    out = np.zeros((4, 2 * n * n), np.float32)
    qdd = np.zeros((4, n), np.float32)
    with pytest.raises(GridError, match="null"):
        lib.forward_dynamics_gradient_device(None, 4, out)
    with pytest.raises(GridError, match="null"):
        lib.forward_dynamics_gradient_device(x, 4, None)
    for stride in (-1, 0, 2 * n, 3 * n - 1):
        with pytest.raises(GridError, match="stride"):
            lib.forward_dynamics_gradient_device(x, 4, out, stride=stride) if stride else lib._check(
                lib.lib.grid_forward_dynamics_gradient_device(lib.handle, ctypes.c_void_p(x.ctypes.data), ctypes.c_int(0), ctypes.c_int(4), ctypes.c_float(9.81),
                                                              ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(0)))
    with pytest.raises(GridError, match="stride"):
        lib.inverse_dynamics_device(x, qdd, 4, np.zeros((4, n), np.float32), stride=n)       # needs q | qd
    with pytest.raises(GridError, match="stride"):
        lib.direct_minv_device(x, 4, np.zeros((4, n * n), np.float32), stride=n - 1)
    with pytest.raises(GridError, match="stride"):
        lib.fdsva_so_device(x, 4, np.zeros((4, 4 * n ** 3), np.float32), stride=2 * n)        # needs q | qd | u
    with pytest.raises(GridError, match="null"):
        lib.forward_dynamics_gradient_qdd_minv_device(x, None, np.zeros((4, n * n), np.float32), 4, out)
    lib.forward_dynamics_gradient_device(None, 0, None)  # an empty batch touches nothing
    lib.forward_dynamics_gradient_device(x, 4, out)
    assert np.isfinite(out).all()
    assert lib.lds_bytes_per_block == (lib.suggested_threads // lib.lanes_per_solve) * (lib.lib.grid_lds_bytes_per_block() // (lib.suggested_threads // lib.lanes_per_solve))
    assert 0 < lib.lds_bytes_per_block <= 160 * 1024


def test_double_precision_and_multi_device_wrappers_check_their_array_shapes():
    """host_f64() and MultiGpuGrid passed arrays of any width on to C entry points that read 3n values per solve."""
    from gridcodegenerator_amd.runtime import MultiGpuGrid

    lib = emu_library("iiwa14", max_timesteps=16)
    n = lib.n
    for alg in ("forward_dynamics", "aba", "idsva_so", "fdsva_so"):
        with pytest.raises(ValueError):
            lib.host_f64(alg, np.zeros((3, 2 * n)))
    with pytest.raises(ValueError):
        lib.host_f64("inverse_dynamics", np.zeros((3, n)))
    with pytest.raises(ValueError):
        lib.host_f64("inverse_dynamics", np.zeros((3, 2 * n)), np.zeros((3, n + 1)))
    with pytest.raises(ValueError):
        lib.host_f64("idsva_so", np.zeros((3, 3 * n)), np.zeros((2, n)))
    assert lib.host_f64("inverse_dynamics", np.zeros((3, 2 * n)), np.zeros((3, n))).shape == (3, n)
    multi = MultiGpuGrid(lib.path, devices=[0, 1], max_timesteps=16)
    try:
        with pytest.raises(ValueError):
            multi.forward_dynamics_gradient_host(np.zeros((4, 2 * n), np.float32))
        with pytest.raises(ValueError):
            multi.forward_dynamics_gradient_host(np.zeros(3 * n, np.float32))
    finally:
        multi.close()

