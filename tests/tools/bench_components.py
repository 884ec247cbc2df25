"""BASELINE.json config 2 (iiwa-14 inverse_dynamics + inverse_dynamics_gradient, batch 1024) and the other stand-alone kernels of
SURVEY.md section 8(f) rows 1-2, each checked against the CPU oracle on the same inputs and timed beside it.  Prints one JSON line per kernel."""
import json, sys, time
sys.path.insert(0, ".")  # run from the repo root
import numpy as np, torch
from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import load
from oracle.rbd_oracle import Oracle

def gpu_time(fn, K=200):
    t_end = time.perf_counter() + 0.05  # clock warm
    while time.perf_counter() < t_end:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / K  # us

def cpu_time(fn, reps=20):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return 1e6 * (time.perf_counter() - t0) / reps

for name, N in (("iiwa14", 1024), ("iiwa14", 16384), ("hyq", 4096), ("atlas", 2048)):
    robot = RobotModel.from_fixture(name); n = robot.n
    lib = load(name, max_timesteps=N)
    orc64, orc32 = Oracle(robot), Oracle(robot, dtype=np.float32)
    rng = np.random.default_rng(0)
    x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
    qdd = rng.uniform(-5, 5, (N, n)).astype(np.float32)
    st = torch.cuda.current_stream().cuda_stream
    d_x, d_qdd = torch.from_numpy(x).cuda(), torch.from_numpy(qdd).cuda()
    d_c = torch.empty((N, n), dtype=torch.float32, device="cuda"); d_M = torch.empty((N, n * n), dtype=torch.float32, device="cuda")
    d_g = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
    def err(got, ref):
        got = got.cpu().numpy().astype(np.float64)
        return float((np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)).max())
    cases = [
        ("inverse_dynamics (with qdd)", lambda: lib.inverse_dynamics_device(d_x, d_qdd, N, d_c, stream=st), d_c, lambda o: o.rnea_batch(x, qdd), 4 * (3 * n + n)),
        ("inverse_dynamics_gradient (with qdd)", lambda: lib.inverse_dynamics_gradient_device(d_x, d_qdd, N, d_g, stream=st), d_g, lambda o: o.rnea_grad_batch(x, qdd), 4 * (3 * n + 2 * n * n)),
        ("direct_minv", lambda: lib.direct_minv_device(d_x, N, d_M, stream=st), d_M, lambda o: o.minv_batch(x), 4 * (n + n * n)),
        ("forward_dynamics", lambda: lib.forward_dynamics_device(d_x, N, d_c, stream=st), d_c, None, 4 * (3 * n + n)),
        ("aba", lambda: lib.aba_device(d_x, N, d_c, stream=st), d_c, None, 4 * (3 * n + n)),
        ("forward_dynamics_gradient", lambda: lib.forward_dynamics_gradient_device(d_x, N, d_g, stream=st), d_g, lambda o: o.fd_grad_batch(x)[0], 4 * (3 * n + 2 * n * n)),
    ]
    gpu_us = [gpu_time(c[1]) for c in cases]  # all GPU timings first: OpenMP workers of the CPU leg spin after a parallel region and starve the launch thread
    for (label, fn, d_out, ref_fn, bytes_per_solve), us in zip(cases, gpu_us):
        line = {"robot": name, "batch": N, "kernel": label, "gpu_us_per_launch": round(us, 2), "gpu_solves_per_s": round(N / us * 1e6),
                "hbm_GBps_algorithmic": round(bytes_per_solve * N / us / 1e3, 1)}
        if ref_fn is not None:
            fn(); torch.cuda.synchronize()
            line["max_rel_err_vs_fp64_oracle"] = err(d_out, ref_fn(orc64))
            cus = cpu_time(lambda: ref_fn(orc32))
            line["cpu_fp32_oracle_solves_per_s"] = round(N / cus * 1e6)
        print(json.dumps(line))
    lib.close()
