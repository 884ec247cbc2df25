"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI (include/grid_capi.h),
against (a) the golden vectors produced by the reference's own NumPy oracle and (b) the CPU oracle on seeded inputs.

Acceptance: per solve max|delta| <= 1e-4 * max|reference| for fp32 kernels vs the fp64 oracle.
"""
import numpy as np
import pytest

from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import build_library, GridLibrary

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROBOTS = ["iiwa14", "hyq", "atlas", "mixed5", "arm6", "chain12", "chain8", "tree12"]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "these tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def libs():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = GridLibrary(build_library(name), device=0, max_timesteps=20000)  # raises when the HIP .so is missing
        return cache[name]

    yield get
    for lib in cache.values():
        lib.close()


def inputs(n, N, seed):
    rng = np.random.default_rng(seed)
    return np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)


def per_solve_err(got, ref):
    got = got.reshape(got.shape[0], -1).astype(np.float64)
    ref = ref.reshape(ref.shape[0], -1)
    return (np.abs(got - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-30)).max()


def run_fd_grad(torch, lib, x):
    N = x.shape[0]
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.full((N, 2 * lib.n * lib.n), float("nan"), dtype=torch.float32, device="cuda")
    lib.forward_dynamics_gradient_device(d_in, N, d_out, stride=x.shape[1], stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


@pytest.mark.parametrize("name", ROBOTS)
def test_fd_grad_matches_reference_goldens(name, torch_cuda, libs, golden):
    g = golden(name)
    lib = libs(name)
    n = lib.n
    x = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)
    out = run_fd_grad(torch_cuda, lib, x)
    ref = np.stack([g["df_du"][k].T.reshape(-1) for k in range(x.shape[0])])  # device layout: [col*n + row]
    assert per_solve_err(out, ref) <= TOL


@pytest.mark.parametrize("name,N", [("iiwa14", 1), ("iiwa14", 37), ("iiwa14", 1024), ("hyq", 4096), ("atlas", 257), ("mixed5", 100), ("arm6", 333), ("chain12", 129), ("chain8", 1000), ("tree12", 515)])
def test_fd_grad_matches_oracle_on_seeded_inputs(name, N, torch_cuda, libs):
    from oracle.rbd_oracle import Oracle

    robot = RobotModel.from_fixture(name)
    lib = libs(name)
    x = inputs(robot.n, N, seed=7)
    out = run_fd_grad(torch_cuda, lib, x)
    ref, _ = Oracle(robot).fd_grad_batch(x.astype(np.float64))
    assert np.isfinite(out).all()
    assert per_solve_err(out, ref) <= TOL


def test_chunked_host_entry_point_gives_the_same_records_as_one_launch(torch_cuda, golden):
    """The hot path's host entry point cuts batches of >= 4096 solves into up to four chunks that travel on the handle's three streams (H2D, kernel and D2H of
    neighbouring chunks overlap) where the caller's buffers are page-locked (grid_host_alloc / GridLibrary.pinned_empty: csrc/grid_capi.hip fd_grad_host).
    Ragged chunk sizes, pageable and page-locked NumPy buffers, bit-identical to a single device-pointer launch; and the same through the single-process multi-handle driver.  (Checked once on the CPU emulation too; it
    is a 10-minute test there.)"""
    from gridcodegenerator_amd.runtime import MultiGpuGrid

    g = golden("iiwa14")
    lib = GridLibrary(build_library("iiwa14"), device=0, max_timesteps=16384)
    multi = MultiGpuGrid(lib.path, devices=[0, 0], max_timesteps=16384)
    n = lib.n
    try:
        for N in (4100, 6151, 16384, 4095):  # two chunks of 2050; three chunks of 2051 / 2051 / 2049; three of 5462 / 5462 / 5460; one launch
            x = np.ascontiguousarray(np.tile(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32), (-(-N // 16), 1))[:N])
            x[:, :n] += np.linspace(0, 0.3, N, dtype=np.float32)[:, None]
            one = run_fd_grad(torch_cuda, lib, x)
            out = lib.forward_dynamics_gradient_host(x)            # pageable buffers: the sequential form
            assert np.isfinite(out).all() and np.array_equal(out, one), N
            xp, op = lib.pinned_empty(x.shape), lib.pinned_empty(one.shape)   # page-locked buffers: the chunked pipeline
            xp[:] = x
            op[:] = np.nan
            assert lib.forward_dynamics_gradient_host(xp, out=op) is op and np.array_equal(op, one), N
            assert np.array_equal(multi.forward_dynamics_gradient_host(x), one), N
            del xp, op
    finally:
        multi.close()
        lib.close()


def test_host_entry_point_and_grid_stride(torch_cuda, libs):
    from oracle.rbd_oracle import Oracle

    robot = RobotModel.from_fixture("iiwa14")
    lib = libs("iiwa14")
    x = inputs(robot.n, 3000, seed=11)
    ref, _ = Oracle(robot).fd_grad_batch(x.astype(np.float64))
    out = lib.forward_dynamics_gradient_host(x)
    assert per_solve_err(out, ref) <= TOL
    for blocks, threads in [(3, 256), (7, 64), (5, 96), (2, 512), (9, 40), (4, 24), (64, 16), (1, 504), (0, 200)]:  # fewer blocks than batches -> grid-stride; ragged block sizes (partial last waves; whole 16-lane rows are used)
        lib.set_launch_dims(blocks, threads)
        out = lib.forward_dynamics_gradient_host(x)
        assert per_solve_err(out, ref) <= TOL, (blocks, threads)
        out = run_fd_grad(torch_cuda, lib, x)  # (fresh NaN-filled output: the host path's device buffer still holds the previous, identical, results)
        assert per_solve_err(out, ref) <= TOL, (blocks, threads)
    lib.set_launch_dims(0, 0)


@pytest.mark.parametrize("name", ["atlas", "tree12"])
def test_branch_frame_path_grid_stride_and_block_sizes(name, torch_cuda, libs):
    """Branched revolute robots run forward_dynamics_gradient on the branch-frame path (own LDS slice, 64-thread blocks by default): every block
    size and a grid smaller than the batch (LDS slices re-used by the grid-stride loop) must give the same results."""
    from oracle.rbd_oracle import Oracle

    robot = RobotModel.from_fixture(name)
    lib = libs(name)
    x = inputs(robot.n, 1500, seed=23)
    ref, _ = Oracle(robot).fd_grad_batch(x.astype(np.float64))
    base = run_fd_grad(torch_cuda, lib, x)
    assert per_solve_err(base, ref) <= TOL
    for blocks, threads in [(0, 256), (11, 64), (5, 128), (3, 512), (0, 32)]:
        lib.set_launch_dims(blocks, threads)
        out = run_fd_grad(torch_cuda, lib, x)
        assert np.array_equal(out, base), (blocks, threads)
    lib.set_launch_dims(0, 0)


@pytest.mark.parametrize("name", ["iiwa14", "atlas"])
def test_full_batch_16384_properties(name, torch_cuda, libs):
    """BASELINE.json's headline size (configs 1 and 4: the 7-DoF arm, the 30-DoF humanoid): size-independent properties instead of a full oracle sweep."""
    from oracle.rbd_oracle import Oracle

    torch = torch_cuda
    robot = RobotModel.from_fixture(name)
    lib = libs(name)
    n, N = robot.n, 16384
    x = inputs(n, N, seed=0)
    out = run_fd_grad(torch, lib, x)
    assert np.isfinite(out).all()
    # (1) a random subset against the oracle
    idx = np.random.default_rng(1).choice(N, 256, replace=False)
    ref, _ = Oracle(robot).fd_grad_batch(x[idx].astype(np.float64))
    assert per_solve_err(out[idx], ref) <= TOL
    # (2) batch-position independence: the same state placed anywhere in the batch gives bit-identical results
    x2 = x.copy()
    x2[5000] = x[3]
    x2[16383] = x[3]
    out2 = run_fd_grad(torch, lib, x2)
    assert np.array_equal(out2[5000], out[3]) and np.array_equal(out2[16383], out[3])
    # (3) determinism
    assert np.array_equal(run_fd_grad(torch, lib, x), out)
    # (4) d qdd / d u_torque is not an output, but d qdd/d qd of a damped joint chain must be finite and the
    #     forward-dynamics gradient must agree with central finite differences of the device forward dynamics
    d_in = torch.from_numpy(x[:64]).cuda()
    eps = 1e-3
    fd = np.zeros((64, 2 * n, n))
    for j in range(2 * n):
        xp, xm = x[:64].copy(), x[:64].copy()
        xp[:, j] += eps
        xm[:, j] -= eps
        qp = torch.empty((64, n), dtype=torch.float32, device="cuda")
        qm = torch.empty((64, n), dtype=torch.float32, device="cuda")
        lib.forward_dynamics_device(torch.from_numpy(xp).cuda(), 64, qp, stream=torch.cuda.current_stream().cuda_stream)
        lib.forward_dynamics_device(torch.from_numpy(xm).cuda(), 64, qm, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        fd[:, j, :] = (qp.cpu().numpy().astype(np.float64) - qm.cpu().numpy()) / (2 * eps)
    got = out[:64].reshape(64, 2 * n, n)
    scale = np.abs(got).max(axis=(1, 2), keepdims=True)
    assert (np.abs(got - fd) / scale).max() < 5e-2  # fp32 finite differences are crude; this catches layout/sign errors


@pytest.mark.parametrize("name", ["iiwa14", "hyq", "atlas", "mixed5", "arm6", "chain12", "chain8", "tree12"])
def test_component_kernels_match_goldens(name, torch_cuda, libs, golden):
    """inverse_dynamics, direct_minv, forward_dynamics, inverse_dynamics_gradient."""
    torch = torch_cuda
    g = golden(name)
    lib = libs(name)
    n = lib.n
    N = g["q"].shape[0]
    st = torch.cuda.current_stream().cuda_stream
    x = torch.from_numpy(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)).cuda()
    qdd = torch.from_numpy(g["qdd"].astype(np.float32)).cuda()
    c = torch.empty((N, n), dtype=torch.float32, device="cuda")
    lib.inverse_dynamics_device(x, None, N, c, stream=st)
    torch.cuda.synchronize()
    assert per_solve_err(c.cpu().numpy(), g["c"]) <= TOL
    lib.inverse_dynamics_device(x, qdd, N, c, stream=st)
    torch.cuda.synchronize()
    assert per_solve_err(c.cpu().numpy(), g["c2"]) <= TOL
    Minv = torch.empty((N, n * n), dtype=torch.float32, device="cuda")
    lib.direct_minv_device(x, N, Minv, stream=st)
    torch.cuda.synchronize()
    ref_upper = np.stack([g["Minv_upper"][k].T.reshape(-1) for k in range(N)])  # column-major, zeros below the diagonal
    assert per_solve_err(Minv.cpu().numpy(), ref_upper) <= TOL
    out_qdd = torch.empty((N, n), dtype=torch.float32, device="cuda")
    lib.forward_dynamics_device(x, N, out_qdd, stream=st)
    torch.cuda.synchronize()
    assert per_solve_err(out_qdd.cpu().numpy(), g["qdd"]) <= TOL
    aba_qdd = torch.empty((N, n), dtype=torch.float32, device="cuda")
    lib.aba_device(x, N, aba_qdd, stream=st)  # articulated-body forward dynamics
    torch.cuda.synchronize()
    assert per_solve_err(aba_qdd.cpu().numpy(), g["qdd"]) <= TOL
    dc = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
    lib.inverse_dynamics_gradient_device(x, qdd, N, dc, stream=st)
    torch.cuda.synchronize()
    ref_dc = np.stack([g["dc_du"][k].T.reshape(-1) for k in range(N)])
    assert per_solve_err(dc.cpu().numpy(), ref_dc) <= TOL
    # the (q,qd,qdd,Minv)-input overload fed with the kernels' own outputs reproduces the u-input result
    df = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
    lib.forward_dynamics_gradient_qdd_minv_device(x, out_qdd, Minv, N, df, stream=st)
    torch.cuda.synchronize()
    ref_df = np.stack([g["df_du"][k].T.reshape(-1) for k in range(N)])
    assert per_solve_err(df.cpu().numpy(), ref_df) <= TOL


def test_single_timing_probe(torch_cuda, libs, golden):
    g = golden("iiwa14")
    lib = libs("iiwa14")
    x = np.hstack([g["q"][5], g["qd"][5], g["u"][5]]).astype(np.float32)
    lib.forward_dynamics_gradient_host(np.zeros((4, 21), np.float32))  # overwrite whatever earlier tests left in the device buffer
    out, us = lib.forward_dynamics_gradient_single_timing(x, reps=100)
    ref = g["df_du"][5].T.reshape(-1)
    assert np.abs(out - ref).max() <= TOL * np.abs(ref).max()
    assert us > 0


@pytest.mark.parametrize("name,threads", [("hyq", 0), ("atlas", 64), ("tree12", 128)])
def test_generated_host_api_float_and_double(name, threads, torch_cuda, golden, tmp_path):
    """Downstream-C++ face of the boundary: a program written against the generated header's host API (init_*, forward_dynamics_gradient<T>,
    close_grid) is compiled with hipcc and run for T=float and T=double; the double instantiation must agree with the fp64 oracle goldens
    to rounding level, which pins the generated ALGORITHM independently of fp32 effects."""
    import os
    import shutil
    import subprocess

    from gridcodegenerator_amd.runtime import HIPCC_FLAGS, generate_header

    g = golden(name)  # (hyq: tip-frame path; atlas, tree12: branch-frame path - in fp64 the formulation itself is pinned to 1e-9;
    n = g["q"].shape[1]  #  the 30-DoF robot needs blocks of 64 threads in double precision: 8 solves per block would exceed the CU's LDS)
    N = g["q"].shape[0]
    gen_dir = tmp_path / "gen"
    generate_header(RobotModel.from_fixture(name), str(gen_dir))
    exe = str(tmp_path / "host_api_demo")
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "host_api_demo.hip")
    subprocess.check_call([shutil.which("hipcc") or "/opt/rocm/bin/hipcc"] + flags + ["-I" + str(gen_dir), src, "-o", exe])
    x = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float64)
    (tmp_path / "in.bin").write_bytes(x.tobytes())
    out = subprocess.check_output([exe, str(tmp_path / "in.bin"), str(N), str(tmp_path / "f32.bin"), str(tmp_path / "f64.bin"), str(threads)], text=True)
    assert "float: overload consistency" in out and "double: overload consistency" in out
    ref = np.stack([g["df_du"][k].T.reshape(-1) for k in range(N)])
    f32 = np.frombuffer((tmp_path / "f32.bin").read_bytes(), dtype=np.float64).reshape(N, -1)
    f64 = np.frombuffer((tmp_path / "f64.bin").read_bytes(), dtype=np.float64).reshape(N, -1)
    assert per_solve_err(f32, ref) <= TOL
    assert per_solve_err(f64, ref) <= 1e-9  # fp64 kernel vs fp64 oracle (inputs are the same doubles)
    for line in out.splitlines():
        assert float(line.split("=")[-1]) <= (1e-4 if line.startswith("float") else 1e-9), line


@pytest.mark.parametrize("name", ["iiwa14", "arm6", "hyq"])
def test_fd_grad_on_ill_conditioned_configurations(name, torch_cuda, libs):
    """Edge cases of the domain: joint angles that line up every other joint axis (the worst conditioned joint-space inertia a chain
    has), a wide input range, and the worst state a 10^6-state random sweep found for the tip-frame path (tests/tools/parity_sweep.py:
    4.5e-5 of max|df/du| near a shoulder + wrist alignment of the 7-DoF arm).  The acceptance bar is the same 1e-4."""
    from oracle.rbd_oracle import Oracle

    robot = RobotModel.from_fixture(name)
    n = robot.n
    lib = libs(name)
    rng = np.random.default_rng(11)
    N = 4096
    x = inputs(n, N, seed=5)
    x[:, 1:n:2] = rng.uniform(-0.02, 0.02, (N, len(range(1, n, 2)))).astype(np.float32)
    x[: N // 8, 1:n:2] = 0.0  # exactly aligned
    wide = np.hstack([rng.uniform(-10 * np.pi, 10 * np.pi, (N, n)), rng.uniform(-10, 10, (N, n)), rng.uniform(-100, 100, (N, n))]).astype(np.float32)
    cases = [x, wide]
    if name == "iiwa14":
        cases.append(np.array([[-1.9726811647415161, -0.15061229467391968, 3.0075197219848633, -1.6245150566101074, 0.2933456301689148, -0.012131131254136562,
                                -0.5525374412536621, -1.6120431423187256, -1.4713057279586792, -1.4227734804153442, 1.8603541851043701, 0.7642495632171631,
                                0.44305285811424255, -1.240814447402954, -6.242157936096191, -5.802548885345459, 8.430603981018066, -4.893815517425537,
                                7.717841625213623, -4.148682594299316, -4.838983535766602]], dtype=np.float32))
    orc = Oracle(robot)
    for xs in cases:
        out = run_fd_grad(torch_cuda, lib, xs)
        ref, _ = orc.fd_grad_batch(xs.astype(np.float64))
        assert np.isfinite(out).all()
        assert per_solve_err(out, ref) <= TOL


@pytest.mark.parametrize("tuning", [{"gradient_walk": "lds"}, {"cols_per_lane": 1}, {"gradient_walk": "registers"}])
def test_non_default_generation_variants_on_gpu(tuning, torch_cuda, golden, tmp_path):
    """The generated forms that the shipped fixtures do not select by default (LDS-assisted forward accumulation for deep trees,
    one derivative column per lane) are exercised on real wave64 hardware too: their LDS hand-offs rely on in-order LDS execution."""
    so = build_library("iiwa14", build_dir=str(tmp_path), tuning=tuning)
    lib = GridLibrary(so, device=0, max_timesteps=4096)
    g = golden("iiwa14")
    x = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)
    out = run_fd_grad(torch_cuda, lib, x)
    ref = np.stack([g["df_du"][k].T.reshape(-1) for k in range(x.shape[0])])
    assert per_solve_err(out, ref) <= TOL
    xs = inputs(7, 3000, seed=3)
    from oracle.rbd_oracle import Oracle

    ref2, _ = Oracle(RobotModel.from_fixture("iiwa14")).fd_grad_batch(xs.astype(np.float64))
    assert per_solve_err(run_fd_grad(torch_cuda, lib, xs), ref2) <= TOL
    lib.close()


@pytest.mark.parametrize("name,tuning", [("tree12", {"branch_walk": "owner"}), ("chain12", {"branch_walk": "path"})])
def test_the_other_walk_of_the_branch_frame_path_on_gpu(name, tuning, torch_cuda, golden, tmp_path):
    """The branch-frame kernels have two forms of their walks along the root paths (algorithms/_branch_frame_gradient.branch_owner_walk): the one a fixture does NOT get by
    default is built here and run on real wave64 hardware - the owner walk's cross-lane reads (ds_bpermute inside 16-lane groups, two hand-downs and two branch crossings on
    the three-level tree) and the path walk on the 12-joint chain - goldens plus 3 000 random states against the fp64 oracle, all first-order kernels of the tree."""
    from oracle.rbd_oracle import Oracle

    so = build_library(name, build_dir=str(tmp_path), tuning=dict({"so_lanes": "off"}, **tuning))
    lib = GridLibrary(so, device=0, max_timesteps=4096)
    g = golden(name)
    n = lib.n
    x = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)
    assert per_solve_err(run_fd_grad(torch_cuda, lib, x), np.stack([g["df_du"][k].T.reshape(-1) for k in range(x.shape[0])])) <= TOL
    xs = inputs(n, 3000, seed=5)
    orc = Oracle(RobotModel.from_fixture(name))
    ref, _ = orc.fd_grad_batch(xs.astype(np.float64))
    assert per_solve_err(run_fd_grad(torch_cuda, lib, xs), ref) <= TOL
    if name == "tree12":  # the stand-alone kernels share the emitter
        torch = torch_cuda
        st = torch.cuda.current_stream().cuda_stream
        N = x.shape[0]
        d_x, d_qdd = torch.from_numpy(x).cuda(), torch.from_numpy(g["qdd"].astype(np.float32)).cuda()
        c = torch.empty((N, n), dtype=torch.float32, device="cuda")
        lib.inverse_dynamics_device(d_x, d_qdd, N, c, stream=st)
        dc = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
        lib.inverse_dynamics_gradient_device(d_x, d_qdd, N, dc, stream=st)
        mi = torch.empty((N, n * n), dtype=torch.float32, device="cuda")
        lib.direct_minv_device(d_x, N, mi, stream=st)
        torch.cuda.synchronize()
        assert per_solve_err(c.cpu().numpy(), g["c2"]) <= TOL
        assert per_solve_err(dc.cpu().numpy(), np.stack([g["dc_du"][k].T.reshape(-1) for k in range(N)])) <= TOL
        assert per_solve_err(mi.cpu().numpy(), np.stack([g["Minv_upper"][k].T.reshape(-1) for k in range(N)])) <= TOL
    lib.close()


@pytest.mark.parametrize("name", ["iiwa14", "arm6", "chain12", "chain8", "hyq", "tree12", "atlas", "mixed5"])
def test_idsva_so_matches_the_restated_reference_algorithm(name, torch_cuda, libs, golden):
    """SURVEY.md section 8(f) rank 3 (serial chains: tip-frame form; the quadruped, the 12-DoF tree and the 30-DoF humanoid: tree form - the humanoid's
    432 KB record goes entry by entry to global memory, GRID_SO_DIRECT): second-order inverse-dynamics derivatives on the GPU vs the NumPy restatement of the
    reference's emitter (oracle/idsva_so_oracle.py - parity unpinned, anchored on finite differences of the pinned first-order oracle)."""
    from gridcodegenerator_amd.robot import DuckRobot
    from oracle.idsva_so_oracle import idsva_so

    torch = torch_cuda
    g = golden(name)
    lib = libs(name)
    n = lib.n
    N = g["q"].shape[0]
    st = torch.cuda.current_stream().cuda_stream
    xh = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)
    qh = g["qdd"].astype(np.float32)
    x, qdd = torch.from_numpy(xh).cuda(), torch.from_numpy(qh).cuda()
    model = DuckRobot(RobotModel.from_fixture(name))
    for use_qdd in (True, False):
        out = torch.full((N, 4 * n ** 3), float("nan"), dtype=torch.float32, device="cuda")
        lib.idsva_so_device(x, qdd if use_qdd else None, N, out, stream=st)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.isfinite(got).all()  # every entry is written exactly once
        for k in range(N):  # every golden state
            ref = idsva_so(model, xh[k, :n].astype(np.float64), xh[k, n:2 * n].astype(np.float64), qh[k].astype(np.float64) if use_qdd else np.zeros(n))
            for t in range(4):
                assert np.abs(got[k].reshape(4, n, n, n)[t] - ref[t]).max() <= TOL * max(np.abs(ref[t]).max(), 1e-3), (k, t)
    # ragged launch: more lane groups than solves, several trips per block
    lib.set_launch_dims(3, 64)
    out2 = torch.full((N, 4 * n ** 3), float("nan"), dtype=torch.float32, device="cuda")
    lib.idsva_so_device(x, None, N, out2, stream=st)
    torch.cuda.synchronize()
    lib.set_launch_dims(0, 0)
    assert np.array_equal(out2.cpu().numpy(), got)


def test_second_order_is_refused_beyond_the_handles_capacity(torch_cuda, libs):
    """The 30-DoF humanoid's records are 432 KB per solve: the generated init_gridData caps the second-order buffers at 1 GiB each and the C ABI rejects
    longer batches instead of overrunning them.  (Robots with prismatic joints have second-order kernels since round 3.)"""
    from gridcodegenerator_amd.runtime import GridError

    torch = torch_cuda
    assert libs("mixed5").has_second_order
    big = GridLibrary(build_library("atlas"), device=0, max_timesteps=4096)
    try:
        assert big.has_second_order
        cap = big.second_order_capacity()
        assert cap == (1 << 30) // (4 * 4 * 30 ** 3) and big.second_order_capacity(f64=True) == cap // 2
        with pytest.raises(GridError):
            big.fdsva_so_host(np.zeros((cap + 1, 90), np.float32))
        with pytest.raises(GridError):
            big.fdsva_so_device(torch.zeros((cap + 1, 90), device="cuda"), cap + 1, torch.zeros((1, 4), device="cuda"))
    finally:
        big.close()


@pytest.mark.parametrize("name", ["iiwa14", "arm6", "chain12", "chain8", "hyq", "tree12", "atlas", "mixed5"])
def test_fdsva_so_matches_the_restated_reference_algorithm(name, torch_cuda, libs, golden):
    """Second half of SURVEY.md section 8(f) rank 3 (serial revolute chains): second-order forward-dynamics derivatives on the GPU vs the NumPy
    restatements of the reference's idsva_so + fdsva_so emitters fed by the pinned first-order oracle (parity unpinned)."""
    from gridcodegenerator_amd.robot import DuckRobot
    from oracle.fdsva_so_oracle import fdsva_so
    from oracle.idsva_so_oracle import idsva_so
    from oracle.rbd_oracle import Oracle

    torch = torch_cuda
    g = golden(name)
    lib = libs(name)
    n = lib.n
    N = g["q"].shape[0]
    st = torch.cuda.current_stream().cuda_stream
    xh = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)
    out = torch.full((N, 4 * n ** 3), float("nan"), dtype=torch.float32, device="cuda")
    lib.fdsva_so_device(torch.from_numpy(xh).cuda(), N, out, stream=st)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    robot = RobotModel.from_fixture(name)
    model, orc = DuckRobot(robot), Oracle(robot)
    for k in range(N):  # every golden state
        q, qd, u = (xh[k, i * n:(i + 1) * n].astype(np.float64) for i in range(3))
        df_du, qdd, Minv, _ = orc.fd_grad(q, qd, u, full=True)
        so = np.concatenate([t.reshape(-1) for t in idsva_so(model, q, qd, qdd)])
        ref = fdsva_so(so, Minv, df_du).reshape(4, n, n, n)
        for t in range(4):
            assert np.abs(got[k].reshape(4, n, n, n)[t] - ref[t]).max() <= TOL * max(np.abs(ref[t]).max(), 1e-3), (k, t)
    # ragged batch on a grid smaller than the batch (grid-stride, partially filled last trip): bit-identical records
    M = N - 3
    lib.set_launch_dims(1, 32)
    out2 = torch.full((M, 4 * n ** 3), float("nan"), dtype=torch.float32, device="cuda")
    lib.fdsva_so_device(torch.from_numpy(xh[:M]).cuda(), M, out2, stream=st)
    torch.cuda.synchronize()
    lib.set_launch_dims(0, 0)
    assert np.array_equal(out2.cpu().numpy(), got[:M])


def _sweep_errors(torch, lib, robot, dist, N, seed):
    from oracle.rbd_oracle import Oracle

    n = robot.n
    aq, aqd, au = {"bench": (np.pi, 2.0, 10.0), "wide": (10 * np.pi, 10.0, 100.0), "aligned": (np.pi, 2.0, 10.0)}[dist]
    orc = Oracle(robot)
    errs = []
    for c in range(0, N, 65536):
        m = min(65536, N - c)
        rng = np.random.default_rng(seed + c)
        x = np.hstack([rng.uniform(-aq, aq, (m, n)), rng.uniform(-aqd, aqd, (m, n)), rng.uniform(-au, au, (m, n))]).astype(np.float32)
        if dist == "aligned":  # consecutive-but-one joint axes (nearly) parallel: the worst conditioned joint-space inertia a chain has
            x[:, 1:n:2] = rng.uniform(-0.02, 0.02, (m, len(range(1, n, 2)))).astype(np.float32)
        out = run_fd_grad(torch, lib, x).astype(np.float64)
        ref, _ = orc.fd_grad_batch(x.astype(np.float64))
        errs.append(np.abs(out - ref).max(axis=1) / np.abs(ref).max(axis=1))
    return np.concatenate(errs)


@pytest.mark.parametrize("name", ["iiwa14", "arm6", "hyq", "atlas", "mixed5"])
def test_non_finite_solves_do_not_leak_into_their_neighbours(name, torch_cuda, libs, golden):
    """ADVICE r2 (see tests/test_generated_emulation.py: same test on the emulation): one diverged trajectory point - NaN in qd, Inf in q, a torque that
    overflows - must not change any other record of the batch (the reference runs one block per solve).  8-lane groups interleave the two solves of a
    16-lane DPP row, so the lane-group scans never touch the other solve's values."""
    torch = torch_cuda
    g = golden(name)
    lib = libs(name)
    n = lib.n
    N = 8
    st = torch.cuda.current_stream().cuda_stream
    x = np.ascontiguousarray(np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float32)[:N])
    qdd = torch.from_numpy(np.ascontiguousarray(g["qdd"].astype(np.float32)[:N])).cuda()
    bad = x.copy()
    bad[2, n + 1] = np.nan
    bad[5, 0] = np.inf
    bad[7, 2 * n] = 1e38
    keep = [k for k in range(N) if k not in (2, 5, 7)]

    def run(inp):
        d = torch.from_numpy(inp).cuda()
        outs = []
        for fn, cols, args in ((lib.forward_dynamics_gradient_device, 2 * n * n, ()), (lib.forward_dynamics_device, n, ()), (lib.aba_device, n, ()), (lib.direct_minv_device, n * n, ())):
            o = torch.zeros((N, cols), dtype=torch.float32, device="cuda")
            if fn == lib.direct_minv_device:
                fn(d, N, o, stream=st)
            else:
                fn(d, *args, N, o, stream=st)
            outs.append(o)
        for fn, cols in ((lib.inverse_dynamics_device, n), (lib.inverse_dynamics_gradient_device, 2 * n * n)):
            o = torch.zeros((N, cols), dtype=torch.float32, device="cuda")
            fn(d, qdd, N, o, stream=st)
            outs.append(o)
        if lib.has_second_order:
            o = torch.zeros((N, 4 * n ** 3), dtype=torch.float32, device="cuda")
            lib.idsva_so_device(d, qdd, N, o, stream=st)
            outs.append(o)
            o = torch.zeros((N, 4 * n ** 3), dtype=torch.float32, device="cuda")
            lib.fdsva_so_device(d, N, o, stream=st)
            outs.append(o)
        torch.cuda.synchronize()
        return [o.cpu().numpy() for o in outs]

    clean, dirty = run(x), run(bad)
    for c, d in zip(clean, dirty):
        assert np.isfinite(c).all()
        assert np.array_equal(c[keep], d[keep])
    assert not np.isfinite(dirty[0][2]).all()


@pytest.mark.parametrize("name,states", [("iiwa14", 2048), ("arm6", 1024), ("hyq", 2048), ("tree12", 2048), ("atlas", 128), ("mixed5", 1024)])
def test_fp32_error_tail_of_the_second_order_kernels_is_guarded(name, states, torch_cuda):
    """the second-order kernels chain the explicit-M factorisation of the first-order path with an n^4 contraction; their fp32 tail had only ever
    been looked at on 12 solves per robot (fdsva_so on the 12-DoF tree: 5.45e-5 of the 1e-4 bar).  This sweep runs idsva_so (random qdd) and fdsva_so on
    `states` random states - half from the bench distribution, half wide (q +-10 pi, qd +-10, u +-100) - and checks EVERY solve against the NumPy
    restatements of the reference's emitters (tests/so_sweep.py; PARITY UNPINNED: the reference holds no vectors for these algorithms).
    Per solve and tensor: max|got - ref| <= 1e-4 max(max|ref|, 1e-3).  tests/tools/parity_sweep_second_order.py prints max / p99.9 of the same sweep."""
    import so_sweep

    robot = RobotModel.from_fixture(name)
    N = states // 2
    lib = GridLibrary(build_library(name), device=0, max_timesteps=N)
    try:
        for dist, seed in (("bench", 7), ("wide", 8)):
            x, qdd = so_sweep.so_inputs(robot.n, N, dist, seed)
            so, f2 = so_sweep.run_so(torch_cuda, lib, x, qdd)
            assert np.isfinite(so).all() and np.isfinite(f2).all()
            e_so, e_f2 = so_sweep.so_errors(name, x, qdd, so, f2)
            assert e_so.max() <= TOL, (name, dist, "idsva_so", e_so.max(), int(e_so.argmax()))
            assert e_f2.max() <= TOL, (name, dist, "fdsva_so", e_f2.max(), int(e_f2.argmax()))
    finally:
        lib.close()


@pytest.mark.parametrize("name,max_tol,p999_tol", [("iiwa14", 4e-5, 1e-5), ("arm6", 3e-5, 1e-5), ("hyq", 3e-5, 1e-5), ("atlas", 6e-5, 1.5e-5)])  # (7-DoF arm: 2.99e-5 worst of 10^6 wide-range states, profiles/r03_parity_sweep.jsonl - the bound leaves a third above it)
def test_fp32_error_tail_of_the_fast_paths_is_guarded(name, max_tol, p999_tol, torch_cuda):
    """the tip-/branch-frame paths form the joint-space inertia explicitly and their fp32 error has a heavier tail than the column walk's
    (round 1: 4.5e-5 worst of 10^6 states on the 7-DoF arm against the 1e-4 acceptance bar, SURVEY.md section 8(c) warning band 2e-5).  This sweep
    (2^18 bench-range states + 2^17 wide-range + 2^17 with every other joint angle within 0.02 rad of zero) keeps the tail from regressing silently:
    serial chains and forests max <= 3e-5 (7-DoF arm 4e-5: its wide-range worst of 10^6 states is 2.99e-5) and 99.9 % <= 1e-5 of max|df/du| (the base-origin family of algorithms/_tip_frame_gradient.py);
    the 30-DoF humanoid (branch frames, factorisation split by branch: 3.6e-5 / 8.8e-6 measured on 2^18 states) max <= 6e-5, 99.9 % <= 1.5e-5."""
    robot = RobotModel.from_fixture(name)
    lib = GridLibrary(build_library(name), device=0, max_timesteps=65536)
    try:
        scale = 1 if robot.n <= 12 else 4  # (the oracle of the 30-DoF robot is 30x slower per state)
        e = np.concatenate([_sweep_errors(torch_cuda, lib, robot, "bench", (1 << 18) // scale, 1000),
                            _sweep_errors(torch_cuda, lib, robot, "wide", (1 << 17) // scale, 2000),
                            _sweep_errors(torch_cuda, lib, robot, "aligned", (1 << 17) // scale, 3000)])
        assert np.isfinite(e).all()
        assert e.max() <= max_tol, (name, e.max())
        assert np.quantile(e, 0.999) <= p999_tol, (name, np.quantile(e, 0.999))
    finally:
        lib.close()


def test_baseline_configs_at_their_full_sizes(torch_cuda, libs):
    """BASELINE.json configs 2-4 at the batch sizes they name (config 1 is the headline test above, config 5 the second-order test below):
    2: iiwa14 inverse_dynamics + inverse_dynamics_gradient, batch 1024 (every solve vs the oracle);
    3: quadruped ABA forward dynamics + forward dynamics + gradient, batch 4096 (every solve of q-dd, a 512-solve subset of the gradient);
    4: 30-DoF humanoid direct_minv + forward_dynamics_gradient, batch 16384 (random subsets)."""
    from oracle.rbd_oracle import Oracle

    torch = torch_cuda
    st = torch.cuda.current_stream().cuda_stream
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    E = lambda *shape: torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")
    # ---- config 2
    robot = RobotModel.from_fixture("iiwa14")
    lib, n, N = libs("iiwa14"), robot.n, 1024
    x, qdd = _so_inputs(n, N, seed=41)
    orc = Oracle(robot)
    c, dc = E(N, n), E(N, 2 * n * n)
    lib.inverse_dynamics_device(T(x), T(qdd), N, c, stream=st)
    lib.inverse_dynamics_gradient_device(T(x), T(qdd), N, dc, stream=st)
    torch.cuda.synchronize()
    x64 = x.astype(np.float64)
    assert per_solve_err(c.cpu().numpy(), orc.rnea_batch(x64, qdd.astype(np.float64))) <= TOL
    assert per_solve_err(dc.cpu().numpy(), orc.rnea_grad_batch(x64, qdd.astype(np.float64))) <= TOL
    # ---- config 3
    robot = RobotModel.from_fixture("hyq")
    lib, n, N = libs("hyq"), robot.n, 4096
    x, _ = _so_inputs(n, N, seed=42)
    orc = Oracle(robot)
    a_aba, a_fd, df = E(N, n), E(N, n), E(N, 2 * n * n)
    lib.aba_device(T(x), N, a_aba, stream=st)
    lib.forward_dynamics_device(T(x), N, a_fd, stream=st)
    lib.forward_dynamics_gradient_device(T(x), N, df, stream=st)
    torch.cuda.synchronize()
    x64 = x.astype(np.float64)
    qdd_ref = np.stack([orc.fd_grad(x64[k, :n], x64[k, n:2 * n], x64[k, 2 * n:], full=True)[1] for k in range(N)])
    assert per_solve_err(a_aba.cpu().numpy(), qdd_ref) <= TOL
    assert per_solve_err(a_fd.cpu().numpy(), qdd_ref) <= TOL
    idx = np.random.default_rng(3).choice(N, 512, replace=False)
    ref, _ = orc.fd_grad_batch(x64[idx])
    assert per_solve_err(df.cpu().numpy()[idx], ref) <= TOL
    # ---- config 4
    robot = RobotModel.from_fixture("atlas")
    n, N = robot.n, 16384
    lib = GridLibrary(build_library("atlas"), device=0, max_timesteps=N)
    try:
        x, _ = _so_inputs(n, N, seed=43)
        orc = Oracle(robot)
        Mi = E(N, n * n)
        lib.direct_minv_device(T(x), N, Mi, stream=st)
        torch.cuda.synchronize()
        got = Mi.cpu().numpy()
        assert np.isfinite(got).all()
        idx = np.random.default_rng(4).choice(N, 128, replace=False)
        ref = orc.minv_batch(x[idx, :n].astype(np.float64))   # device layout: column-major, upper triangle (zeros below the diagonal)
        assert per_solve_err(got[idx], ref) <= TOL
    finally:
        lib.close()


@pytest.mark.parametrize("name", ["iiwa14", "hyq", "atlas"])
def test_double_precision_c_abi_on_the_gpu(name, torch_cuda, golden):
    """The *_f64 entry points (T = double instantiations of the same generated kernels) against the goldens of the reference's oracle to rounding
    level, on the hardware: tip-frame path (arm, quadruped) and branch-frame path (humanoid: the library must launch fewer solves per block in double
    precision so that a block fits the LDS of a CU)."""
    g = golden(name)
    lib = GridLibrary(build_library(name), device=0, max_timesteps=64)
    try:
        n, N = lib.n, g["q"].shape[0]
        x = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float64)
        col = lambda key: np.stack([g[key][k].T.reshape(-1) for k in range(N)])
        assert per_solve_err(lib.forward_dynamics_gradient_host_f64(x), col("df_du")) <= 1e-9
        torch = torch_cuda
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.full((N, 2 * n * n), float("nan"), dtype=torch.float64, device="cuda")
        lib.forward_dynamics_gradient_device_f64(d_in, N, d_out, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert per_solve_err(d_out.cpu().numpy(), col("df_du")) <= 1e-9
        assert per_solve_err(lib.host_f64("inverse_dynamics", x[:, :2 * n], g["qdd"]), g["c2"]) <= 1e-9
        assert per_solve_err(lib.host_f64("inverse_dynamics_gradient", x[:, :2 * n], g["qdd"]), col("dc_du")) <= 1e-9
        assert per_solve_err(lib.host_f64("direct_minv", x[:, :n]), col("Minv_upper")) <= 1e-9
        assert per_solve_err(lib.host_f64("forward_dynamics", x), g["qdd"]) <= 1e-9
        assert per_solve_err(lib.host_f64("aba", x), g["qdd"]) <= 1e-9
        out32 = lib.forward_dynamics_gradient_host(x.astype(np.float32))  # float and double state of one handle side by side
        assert per_solve_err(out32, col("df_du")) <= TOL
    finally:
        lib.close()


@pytest.mark.parametrize("name", ["hyq", "atlas"])
def test_double_precision_second_order_on_branched_robots(name, torch_cuda, golden):
    """idsva_so / fdsva_so of the tree form in T = double through the C ABI, to rounding level against the restated reference algorithm (the quadruped
    stages its records in LDS, the humanoid writes them straight to global memory and reads the idsva_so tensors back from the handle's workspace)."""
    from gridcodegenerator_amd.robot import DuckRobot
    from oracle.fdsva_so_oracle import fdsva_so
    from oracle.idsva_so_oracle import idsva_so
    from oracle.rbd_oracle import Oracle

    g = golden(name)
    lib = GridLibrary(build_library(name), device=0, max_timesteps=16)
    try:
        n, N = lib.n, 4
        x = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float64)[:N]
        qdd_in = g["qdd"][:N].astype(np.float64)
        robot = RobotModel.from_fixture(name)
        model, orc = DuckRobot(robot), Oracle(robot)
        got_id = lib.host_f64("idsva_so", x, qdd_in)
        got_fd = lib.host_f64("fdsva_so", x)
        for k in range(N):
            q, qd, u = (x[k, i * n:(i + 1) * n] for i in range(3))
            ref = np.concatenate([t.reshape(-1) for t in idsva_so(model, q, qd, qdd_in[k])])
            assert np.abs(got_id[k] - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1.0), k
            df_du, qdd, Minv, _ = orc.fd_grad(q, qd, u, full=True)
            so = np.concatenate([t.reshape(-1) for t in idsva_so(model, q, qd, qdd)])
            ref2 = fdsva_so(so, Minv, df_du).reshape(-1)
            assert np.abs(got_fd[k] - ref2).max() <= 1e-9 * max(np.abs(ref2).max(), 1.0), k
    finally:
        lib.close()


def test_single_process_multi_handle_driver_on_the_gpu(torch_cuda):
    """SURVEY.md section 8(e) in one process: G handles (all on the one GPU of the test box, which is how the split is rehearsed without a
    multi-GPU node), one host thread per handle, the batch cut into contiguous ranges - bit-identical to the one-handle result."""
    from gridcodegenerator_amd.runtime import MultiGpuGrid

    so = build_library("iiwa14")
    single = GridLibrary(so, device=0, max_timesteps=16384)
    x = inputs(7, 16384, seed=0)
    ref = single.forward_dynamics_gradient_host(x)
    for G in (2, 3, 8):
        multi = MultiGpuGrid(so, devices=[0] * G, max_timesteps=16384)
        try:
            assert np.array_equal(multi.forward_dynamics_gradient_host(x), ref), G
            assert np.array_equal(multi.forward_dynamics_gradient_host(x[:1001]), ref[:1001]), G  # ragged split
        finally:
            multi.close()
    single.close()


def _so_inputs(n, N, seed):
    rng = np.random.default_rng(seed)
    x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
    qdd = rng.uniform(-5, 5, (N, n)).astype(np.float32)
    return x, qdd


def test_second_order_kernels_at_config5_batch_65536(torch_cuda):
    """BASELINE.json config 5 (iiwa-14 idsva_so / fdsva_so, batch 65536) under real test: after other kernels have dirtied the LDS, NaN-prefilled
    outputs, EVERY solve of a 256-solve random subset against the restated oracles (parity unpinned: the reference holds no vectors for the
    second-order algorithms; oracle/idsva_so_oracle.py restates its emitter and is anchored on finite differences of the pinned first-order
    oracle), batch-position independence and determinism bit-exact."""
    from gridcodegenerator_amd.robot import DuckRobot
    from oracle.fdsva_so_oracle import fdsva_so
    from oracle.idsva_so_oracle import idsva_so
    from oracle.rbd_oracle import Oracle

    torch = torch_cuda
    robot = RobotModel.from_fixture("iiwa14")
    n, N = robot.n, 65536
    lib = GridLibrary(build_library("iiwa14"), device=0, max_timesteps=N)
    try:
        st = torch.cuda.current_stream().cuda_stream
        x, qdd = _so_inputs(n, N, seed=2024)
        x[40000] = x[3]; qdd[40000] = qdd[3]      # the same state at three positions of the batch (first block, middle, last lane group)
        x[N - 1] = x[3]; qdd[N - 1] = qdd[3]
        d_x, d_qdd = torch.from_numpy(x).cuda(), torch.from_numpy(qdd).cuda()
        junk = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
        lib.forward_dynamics_gradient_device(d_x, N, junk, stream=st)            # leaves arbitrary bit patterns in LDS
        lib.inverse_dynamics_gradient_device(d_x, d_qdd, N, junk, stream=st)

        def run_id():
            out = torch.full((N, 4 * n ** 3), float("nan"), dtype=torch.float32, device="cuda")
            lib.idsva_so_device(d_x, d_qdd, N, out, stream=st)
            torch.cuda.synchronize()
            return out.cpu().numpy()

        def run_fd():
            out = torch.full((N, 4 * n ** 3), float("nan"), dtype=torch.float32, device="cuda")
            lib.fdsva_so_device(d_x, N, out, stream=st)
            torch.cuda.synchronize()
            return out.cpu().numpy()

        so, df2 = run_id(), run_fd()
        assert np.isfinite(so).all() and np.isfinite(df2).all()                    # every one of the 4 n^3 entries of every solve is written
        for got in (so, df2):
            assert np.array_equal(got[40000], got[3]) and np.array_equal(got[N - 1], got[3])   # batch-position independence
        assert np.array_equal(run_id(), so) and np.array_equal(run_fd(), df2)     # determinism
        model, orc = DuckRobot(robot), Oracle(robot)
        idx = np.random.default_rng(9).choice(N, 256, replace=False)
        worst = [0.0, 0.0]
        for k in idx:
            q, qd, u = (x[k, i * n:(i + 1) * n].astype(np.float64) for i in range(3))
            ref = np.stack([t for t in idsva_so(model, q, qd, qdd[k].astype(np.float64))]).reshape(4, -1)
            g4 = so[k].reshape(4, -1)
            worst[0] = max(worst[0], max(np.abs(g4[t] - ref[t]).max() / max(np.abs(ref[t]).max(), 1e-3) for t in range(4)))
            df_du, qdd_fd, Minv, _ = orc.fd_grad(q, qd, u, full=True)
            so_fd = np.concatenate([t.reshape(-1) for t in idsva_so(model, q, qd, qdd_fd)])
            ref2 = fdsva_so(so_fd, Minv, df_du).reshape(4, -1)
            g4 = df2[k].reshape(4, -1)
            worst[1] = max(worst[1], max(np.abs(g4[t] - ref2[t]).max() / max(np.abs(ref2[t]).max(), 1e-3) for t in range(4)))
        assert worst[0] <= TOL and worst[1] <= TOL, worst
    finally:
        lib.close()


def test_second_order_tensor_layout_from_gpu_outputs_only(torch_cuda, libs):
    """Pins the [i][j][k] index layout of the 4 n^3 records (reference algorithms/_idsva_so.py:156-159,583-586; _fdsva_so.py:74-81) WITHOUT the
    restated oracle: central differences of the GPU's own first-order kernels (see tests/so_layout_check.py)."""
    from so_layout_check import check_second_order_layout

    torch = torch_cuda
    st = torch.cuda.current_stream().cuda_stream

    class Dev:
        stream = st
        arr = staticmethod(lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda())
        full = staticmethod(lambda shape, v: torch.full(shape, v, dtype=torch.float32, device="cuda"))

        @staticmethod
        def host(t):
            torch.cuda.synchronize()
            return t.cpu().numpy()

    check_second_order_layout(libs("iiwa14"), Dev, B=6)


def test_large_batch_component_kernels_after_lds_reuse(torch_cuda, libs):
    """Regression: lanes of a lane group that hold no joint once read uninitialised LDS as their qdd; 0 * NaN then leaked into every joint's
    composite sums.  It only showed with large batches (LDS re-used by many blocks): 65536 solves of ID and ID-gradient after other kernels ran."""
    from oracle.rbd_oracle import Oracle

    torch = torch_cuda
    robot = RobotModel.from_fixture("iiwa14")
    n, N = robot.n, 65536
    lib = GridLibrary(build_library("iiwa14"), device=0, max_timesteps=N)
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(77)
    x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
    qdd = rng.uniform(-5, 5, (N, n)).astype(np.float32)
    d_x, d_qdd = torch.from_numpy(x).cuda(), torch.from_numpy(qdd).cuda()
    junk = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
    lib.forward_dynamics_gradient_device(d_x, N, junk, stream=st)  # leaves arbitrary bit patterns in LDS
    c = torch.empty((N, n), dtype=torch.float32, device="cuda")
    lib.inverse_dynamics_device(d_x, d_qdd, N, c, stream=st)
    g = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
    lib.inverse_dynamics_gradient_device(d_x, d_qdd, N, g, stream=st)
    torch.cuda.synchronize()
    orc = Oracle(robot)
    assert per_solve_err(c.cpu().numpy(), orc.rnea_batch(x.astype(np.float64), qdd.astype(np.float64))) <= TOL
    assert per_solve_err(g.cpu().numpy(), orc.rnea_grad_batch(x.astype(np.float64), qdd.astype(np.float64))) <= TOL
    lib.close()


@pytest.mark.parametrize("name,so_threads_f64", [("iiwa14", 0), ("hyq", 32)])
def test_generated_host_api_first_and_second_order_float_and_double(name, so_threads_f64, torch_cuda, golden, tmp_path):
    """The generated host API of a serial-chain robot (tip-frame path, chain form of the second order) and of a branched one (a forest: tip-frame path per limb, tree
    form of the second order) for T = float and T = double: forward_dynamics_gradient<T>, forward_dynamics<T>, idsva_so_host<T, true>, fdsva_so<T>.
    The double instantiations must agree with the fp64 oracles to rounding level - that pins the generated algorithms (incl. the DPP scans on 64-bit
    values, the register factorisation, the level-by-level tree propagation) independently of fp32 effects."""
    import os
    import shutil
    import subprocess

    from gridcodegenerator_amd.robot import DuckRobot
    from gridcodegenerator_amd.runtime import HIPCC_FLAGS, generate_header
    from oracle.fdsva_so_oracle import fdsva_so
    from oracle.idsva_so_oracle import idsva_so
    from oracle.rbd_oracle import Oracle

    g = golden(name)
    n, N = g["q"].shape[1], 4
    gen_dir = tmp_path / "gen"
    generate_header(RobotModel.from_fixture(name), str(gen_dir))
    exe = str(tmp_path / "host_api_so_demo")
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "host_api_so_demo.hip")
    subprocess.check_call([shutil.which("hipcc") or "/opt/rocm/bin/hipcc"] + flags + ["-I" + str(gen_dir), src, "-o", exe])
    x = np.hstack([g["q"], g["qd"], g["u"]]).astype(np.float64)[:N]
    (tmp_path / "in.bin").write_bytes(x.tobytes())
    out = subprocess.check_output([exe, str(tmp_path / "in.bin"), str(N), str(tmp_path / "o"), str(so_threads_f64)], text=True)
    assert "done" in out
    assert out.count("with SUGGESTED_THREADS: mismatches = 0") == 4, out  # (the second-order hosts launched with the general block size)
    robot = RobotModel.from_fixture(name)
    model, orc = DuckRobot(robot), Oracle(robot)
    load = lambda tag, what: np.frombuffer((tmp_path / ("o_%s_%s.bin" % (tag, what))).read_bytes(), dtype=np.float64).reshape(N, -1)
    for tag, tol in (("f32", TOL), ("f64", 1e-9)):
        dfdu, so, df2 = load(tag, "dfdu"), load(tag, "so"), load(tag, "df2")
        assert per_solve_err(dfdu, np.stack([g["df_du"][k].T.reshape(-1) for k in range(N)])) <= tol
        for k in range(N):
            q, qd, u = (x[k, i * n:(i + 1) * n] for i in range(3))
            df_du, qdd, Minv, _ = orc.fd_grad(q, qd, u, full=True)
            ref_so = np.concatenate([t.reshape(-1) for t in idsva_so(model, q, qd, qdd)])
            ref_df2 = fdsva_so(ref_so, Minv, df_du)
            for got, ref in ((so[k], ref_so), (df2[k], ref_df2)):
                for t in range(4):
                    a, b = got.reshape(4, -1)[t], ref.reshape(4, -1)[t]
                    assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-3), (tag, k, t)


@pytest.mark.parametrize("seed,n", [(16, 8), (2, 14), (4, 27), (58, 20)])
def test_random_trees_on_the_gpu(seed, n, torch_cuda, tmp_path):
    """Generator robustness on the real hardware: random tree topologies (three tree levels, 16- and 32-lane groups, branch hand-over
    records inside / outside the X(q) storage; seed 58: the owner walk on a three-level tree with root paths of 11 joints) are generated,
    compiled for gfx950 here and checked against the oracle, every kernel."""
    from oracle.rbd_oracle import Oracle
    from test_generated_emulation import _random_tree_description

    robot = RobotModel(_random_tree_description(seed, n))
    lib = GridLibrary(build_library(robot, build_dir=str(tmp_path)), device=0, max_timesteps=2048)
    try:
        orc = Oracle(robot)
        N = 700
        x = inputs(n, N, seed=seed)
        ref, _ = orc.fd_grad_batch(x.astype(np.float64))
        for blocks, threads in [(0, 0), (7, 128)]:
            lib.set_launch_dims(blocks, threads)
            assert per_solve_err(run_fd_grad(torch_cuda, lib, x), ref) <= TOL, (blocks, threads)
        lib.set_launch_dims(0, 0)
        torch = torch_cuda
        st = torch.cuda.current_stream().cuda_stream
        qdd = np.random.default_rng(seed).uniform(-5, 5, (N, n)).astype(np.float32)
        d_x, d_qdd = torch.from_numpy(x).cuda(), torch.from_numpy(qdd).cuda()
        d_c = torch.full((N, n), float("nan"), dtype=torch.float32, device="cuda")
        lib.inverse_dynamics_device(d_x, d_qdd, N, d_c, stream=st)
        d_dc = torch.full((N, 2 * n * n), float("nan"), dtype=torch.float32, device="cuda")
        lib.inverse_dynamics_gradient_device(d_x, d_qdd, N, d_dc, stream=st)
        d_acc = torch.full((N, n), float("nan"), dtype=torch.float32, device="cuda")
        lib.forward_dynamics_device(d_x, N, d_acc, stream=st)
        torch.cuda.synchronize()
        x64 = x.astype(np.float64)
        assert per_solve_err(d_c.cpu().numpy(), orc.rnea_batch(x64[:, :2 * n], qdd.astype(np.float64))) <= TOL
        assert per_solve_err(d_dc.cpu().numpy(), orc.rnea_grad_batch(x64[:, :2 * n], qdd.astype(np.float64))) <= TOL
        qdd_ref = np.stack([orc.fd_grad(x64[k, :n], x64[k, n:2 * n], x64[k, 2 * n:], full=True)[1] for k in range(64)])
        assert per_solve_err(d_acc.cpu().numpy()[:64], qdd_ref) <= TOL
    finally:
        lib.close()
