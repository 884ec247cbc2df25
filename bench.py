#!/usr/bin/env python3
"""Headline benchmark: forward_dynamics_gradient solves/sec, iiwa-14, batch 16384 (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path (one forward_dynamics_gradient_kernel launch through the C ABI) over the job's batch of
synthetic solves whose inputs and outputs are resident in HBM.

Multi-GPU: the batch axis shards with no collective.
  --scaling strong (default): ONE global batch of 16384 solves, rank r owns the contiguous range [r*B/G, (r+1)*B/G) of it;
  --scaling weak: every rank owns its own 16384-solve batch.  With N > 1 the line also carries the other mode as a second key.
value = solves processed by all ranks / max-over-ranks wall time of the K steps.  Rank 0 prints ONE JSON line.

Protocol of the timed region: `--warmup` untimed steps, barrier + synchronize, EXACTLY `--steps` launches bracketed by one HIP event
pair on the launch stream, synchronize, barrier.  Before the warm-up steps a DISCLOSED, untimed clock warm ("clock_warm_ms") keeps the
GPU busy so that a short run is not timed on a clock still ramping up from idle.  A second, diagnostic pass of K launches with one
event every 4 launches gives the per-launch distribution ("launch_us_median"); it is not part of `value`.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

ROBOT = "iiwa14"
BATCH = 16384
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s measured copy ceiling
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector
FLOPS_PER_SOLVE_REFERENCE = 53e3   # static estimate of the REFERENCE's emitted code, n = 7 (not what this kernel executes)
N_SIMDS = 1024                 # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4                # MI355X_MICROARCH.md: max clock
PMC_FILE = os.path.join("profiles", "pmc_counters.json")   # per-launch counters of the headline kernel from committed rocprofv3 --pmc passes


def make_inputs(n, N, seed=0):
    """default_rng(0); q~U(-pi,pi), qd~U(-2,2), u~U(-10,10); float32 AoS [k][3n]."""
    rng = np.random.default_rng(seed)
    return np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)


def shard_range(N, world, rank):
    """Strong scaling: contiguous range [k0, k1) of the global batch owned by `rank` (same arithmetic as csrc/grid_capi.hip: multi_range)."""
    per = (N + world - 1) // world
    return min(rank * per, N), min((rank + 1) * per, N)


def rank_inputs(n, N, world, rank, scaling):
    """This rank's synthetic shard.  strong: its slice of the ONE global seeded batch; weak: its own batch (seed = rank)."""
    if scaling == "strong":
        k0, k1 = shard_range(N, world, rank)
        return make_inputs(n, N, seed=0)[k0:k1]
    return make_inputs(n, N, seed=rank)


def _timed_rate(fn, nsolves, budget_s, max_reps=100000):
    reps = 0
    t0 = time.perf_counter()
    while True:
        fn()
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or reps >= max_reps:
            break
    return reps * nsolves / dt, reps, dt


def cpu_baseline(robot, x, budget_s=10.0):
    """Times the CPU oracle (oracle/rbd_oracle.c, the C restatement of the reference's NumPy oracle) on this host's cores: the checker
    used as a reported baseline, never as part of the measured GPU path.  ~20 s of CPU work in total: thread-count sweep, the all-cores
    fp32 sample, and single-thread fp32 / fp64 samples."""
    from oracle import rbd_oracle

    so = None
    try:  # rebuild for this host's ISA outside the tree (the in-tree .so is the portable build)
        so = rbd_oracle.build(march="native", out=os.path.join("/tmp", "librbd_oracle_native_%d.so" % os.getpid()), force=True)
    except Exception:
        so = None
    orc = rbd_oracle.Oracle(robot, dtype=np.float32, lib_path=so)
    orc64 = rbd_oracle.Oracle(robot, dtype=np.float64, lib_path=so)
    sample = x  # the whole bench batch per call (enough work per thread for the static OpenMP split)
    # the job may own fewer CPUs than the host shows (a 1-GPU box gets a share of the host): pick the thread count that is
    # actually fastest from a short sweep, then time the bounded sample with it; `cores` reports the threads used
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    best_nt, best_rate = 1, 0.0
    for nt in sorted(set(min(avail, c) for c in (4, 8, 16, 32, 64, 128, avail))):
        orc.fd_grad_batch(sample, nthreads=nt)
        rate, _, _ = _timed_rate(lambda: orc.fd_grad_batch(sample, nthreads=nt), 1, 0.4)
        if rate > best_rate:
            best_nt, best_rate = nt, rate
    cores = best_nt
    value, reps, dt = _timed_rate(lambda: orc.fd_grad_batch(sample, nthreads=cores), sample.shape[0], budget_s)
    small = sample[:2048]
    small64 = small.astype(np.float64)
    r32, _, _ = _timed_rate(lambda: orc.fd_grad_batch(small, nthreads=1), small.shape[0], 2.0)
    r64, _, _ = _timed_rate(lambda: orc64.fd_grad_batch(small64, nthreads=1), small.shape[0], 2.0)
    return {"value": value, "unit": "solves/s", "cores": int(cores), "kind": "port",
            "sample": "%d x %d iiwa14 solves of the bench batch, fp32 C oracle (oracle/rbd_oracle.c, -O3 -march=native, OpenMP static, best of a thread-count sweep on %d visible CPUs), %.1f s" % (reps, sample.shape[0], avail, dt),
            "single_thread_fp32": r32, "single_thread_fp64": r64, "single_thread_sample": "2048 solves per call, 2 s each"}


def reduce_max(elapsed, dist, device):
    import torch

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_min_max(elapsed, dist, device):
    """(min, max) of the ranks' elapsed times: max is the job's time, min shows how far the fastest rank was ahead (a straggler in a scaling curve)."""
    import torch

    t = torch.tensor([-elapsed, elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(-t[0].item()), float(t[1].item())


def harness_selftest(args):
    """Multi-rank plumbing only (gloo, CPU): no kernels, no metric.  Checks that the ranks' shards are what the scaling mode defines:
    strong -> the concatenation of the ranks' slices IS the one global batch; weak -> pairwise distinct batches."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    n, N = 7, args.batch
    x = rank_inputs(n, N, world, rank, args.scaling)
    k0, k1 = shard_range(N, world, rank) if args.scaling == "strong" else (0, N)
    info = torch.tensor([float(k0), float(k1), float(x.shape[0]), float(np.abs(x.astype(np.float64)).sum())], dtype=torch.float64)
    infos = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(infos, info)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))  # rank-dependent "work": the reported time must be the slowest rank's
    fastest, elapsed = reduce_min_max(time.perf_counter() - t0, dist, torch.device("cpu"))
    dist.barrier()
    if rank == 0:
        ranges = [(int(i[0]), int(i[1])) for i in infos]
        total_checksum = float(sum(i[3] for i in infos))
        global_checksum = float(np.abs(make_inputs(n, N, seed=0).astype(np.float64)).sum())
        print(json.dumps({"selftest": "bench harness", "n_gpus": world, "scaling": args.scaling, "max_elapsed_s": elapsed, "per_rank_ms": {"min": 1e3 * fastest, "max": 1e3 * elapsed},
                          "ranges": ranges, "rows": [int(i[2]) for i in infos],
                          "contiguous_cover": args.scaling == "strong" and ranges[0][0] == 0 and ranges[-1][1] == N and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1)),
                          "shards_sum_to_global_batch": abs(total_checksum - global_checksum) <= 1e-9 * global_checksum,
                          "distinct_shards": len(set(round(float(i[3]), 6) for i in infos)) == world,
                          "solves_counted": (N if args.scaling == "strong" else world * N) * args.steps}))
    dist.destroy_process_group()


def measured_copy_bandwidth(torch, dev):
    """HBM bandwidth a plain device-to-device copy reaches on this GPU (read + write bytes / time): the measured ceiling next to the 8 TB/s spec."""
    nbytes = 1 << 30
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    del a, b
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def other_configs(torch):
    """BASELINE.json's secondary configs on this GPU, device-resident, HIP events around 20 launches after 0.15 s of warm-up launches each - diagnostics next to the headline,
    never `value`: config 3 (12-DoF quadruped @4 096), config 4 (30-DoF humanoid @16 384 on ONE GPU), config 5 (second order, 7-DoF arm @65 536 on ONE GPU)."""
    from gridcodegenerator_amd import RobotModel
    from gridcodegenerator_amd.runtime import load

    out = {}
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(f):
        t_end = time.perf_counter() + 0.15  # (clock warm, as before the headline's timed region)
        while time.perf_counter() < t_end:
            for _ in range(10):
                f()
            torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        return round(1e3 * e0.elapsed_time(e1) / 20, 2)

    for name, N, second_order in (("hyq", 4096, False), ("atlas", 16384, False), ("iiwa14", 65536, True)):
        n = RobotModel.from_fixture(name).n
        lib = load(name, max_timesteps=N)
        rng = np.random.default_rng(0)
        x = torch.from_numpy(np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)).cuda()
        r = {}
        if second_order:
            qdd = torch.from_numpy(rng.uniform(-5, 5, (N, n)).astype(np.float32)).cuda()
            so = torch.empty((N, 4 * n ** 3), dtype=torch.float32, device="cuda")
            by = 4 * (3 * n + 4 * n ** 3) * N
            for kern, f in (("idsva_so", lambda: lib.idsva_so_device(x, qdd, N, so, stream=st)), ("fdsva_so", lambda: lib.fdsva_so_device(x, N, so, stream=st))):
                us = timed(f)
                r[kern] = {"us_per_launch": us, "solves_per_s": round(N / us * 1e6), "hbm_frac_of_8TBps": round(by / us / 1e3 / 8000, 3)}
            del so, qdd
        else:
            o = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
            us = timed(lambda: lib.forward_dynamics_gradient_device(x, N, o, stream=st))
            r["forward_dynamics_gradient"] = {"us_per_launch": us, "solves_per_s": round(N / us * 1e6), "hbm_frac_of_8TBps": round(4 * (3 * n + 2 * n * n) * N / us / 1e3 / 8000, 3)}
            if name == "atlas":
                mi = torch.empty((N, n * n), dtype=torch.float32, device="cuda")
                us = timed(lambda: lib.direct_minv_device(x, N, mi, stream=st))
                r["direct_minv"] = {"us_per_launch": us, "solves_per_s": round(N / us * 1e6)}
            else:
                q = torch.empty((N, n), dtype=torch.float32, device="cuda")
                us = timed(lambda: lib.aba_device(x, N, q, stream=st))
                r["aba"] = {"us_per_launch": us, "solves_per_s": round(N / us * 1e6)}
        out["%s @%d" % (name, N)] = r
        lib.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=BATCH, help="global batch (strong scaling) / solves per GPU (weak scaling) per step")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (default): one global batch of --batch solves cut into contiguous per-GPU ranges; weak: --batch solves per GPU")
    ap.add_argument("--clock-warm-ms", type=float, default=50.0, help="untimed, disclosed GPU clock warm before the warm-up steps (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the copy-bandwidth, end-to-end and second-scaling-mode measurements")
    ap.add_argument("--threads", type=int, default=0, help="threads per block override (0 = library default)")
    ap.add_argument("--blocks", type=int, default=0, help="blocks override (0 = one lane-group batch per block)")
    ap.add_argument("--build-dir", default=None, help="load the robot library from another build directory (tuning experiments)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle spot check (timing ablation builds produce wrong results on purpose)")
    ap.add_argument("--dump", default=None, help="directory: every rank saves its output shard as out_rank<r>.npy (tests: the shards of a strong-scaling job concatenate to the 1-rank result)")
    ap.add_argument("--harness-selftest", action="store_true",
                    help="CPU-only check of the multi-rank harness (gloo rendezvous, sharding, barrier, MAX-reduce, single JSON line); "
                         "runs NO dynamics and reports NO metric - used by tests/test_bench_distributed.py")
    args = ap.parse_args()
    if args.harness_selftest:
        return harness_selftest(args)

    import torch

    from gridcodegenerator_amd import RobotModel
    from gridcodegenerator_amd.runtime import load

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("GRID_BENCH_FORCE_DIST", "0") == "1"  # (FORCE_DIST: tests only - the RCCL code path with a single rank)
    rehearsal = False
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU over RCCL.  GRID_BENCH_REHEARSAL=1 (tests only) runs the same code path with gloo and lets ranks share a GPU,
        # so that sharding, barriers and the MAX-reduce can be exercised on a one-GPU box
        rehearsal = os.environ.get("GRID_BENCH_REHEARSAL", "0") == "1"
        if rehearsal:
            local_rank = local_rank % max(1, torch.cuda.device_count())
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    red_dev = torch.device("cpu") if (distributed and rehearsal) else dev

    robot = RobotModel.from_fixture(ROBOT)
    n = robot.n
    B = args.batch
    lib = load(ROBOT, device=local_rank, max_timesteps=B, build_dir=args.build_dir)  # raises if the HIP library is missing (no CPU fallback)
    if args.threads or args.blocks:
        lib.set_launch_dims(args.blocks, args.threads)
    stream = torch.cuda.current_stream(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def run_mode(scaling, steps, warmup, clock_warm_ms):
        """One complete measurement in one scaling mode; returns a dict of this job's numbers (rank 0's view after the MAX-reduce)."""
        x = rank_inputs(n, B, world, rank, scaling)   # strong: this rank's contiguous slice of the ONE global batch
        N = x.shape[0]
        d_in = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        d_out = torch.empty((max(N, 1), 2 * n * n), dtype=torch.float32, device=dev)
        step = lib.prepare_forward_dynamics_gradient_device(d_in, N, d_out, stream=stream.cuda_stream)  # ctypes arguments built once
        warm_launches = 0
        gc.collect()   # (before the clock warm, not next to the timed region: a collection there idles the GPU for tens of ms and the clocks fall back)
        gc.disable()
        if clock_warm_ms > 0:  # disclosed, untimed: keep the GPU busy so that the clocks have ramped up before anything is timed
            t_end = time.perf_counter() + clock_warm_ms * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(64):
                    step()
                warm_launches += 64
                torch.cuda.synchronize(dev)
        # the W warm-up steps run through the SAME code path as the timed region (events recorded on the launch stream, synchronize):
        # first-use costs of the event machinery must not land in a 20-step timed region
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(warmup):
            step()
        ev1.record(stream)
        torch.cuda.synchronize(dev)
        ev0.elapsed_time(ev1)
        barrier()
        # ---- timed region: exactly K steps; ONE HIP event pair on the launch stream brackets it
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(steps):
            step()
        ev1.record(stream)
        torch.cuda.synchronize(dev)                 # this rank's K steps are complete ...
        elapsed = time.perf_counter() - t0           # ... so its clock stops here; the closing barrier below is not part of the workload
        gc.enable()
        barrier()
        gpu_ms = ev0.elapsed_time(ev1)
        fastest = elapsed
        if distributed:
            fastest, elapsed = reduce_min_max(elapsed, dist, red_dev)  # job time = slowest rank (all ranks started together behind the opening barrier)
        # ---- diagnostic pass (not part of `value`): an event every CHUNK launches -> distribution of the per-launch time.  (An event after
        # every single launch puts a marker packet between the kernels that costs ~3 us, more than a third of the kernel itself.)
        CHUNK = 4
        nch = max(1, steps // CHUNK)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nch + 1)]
        evs[0].record(stream)
        for i in range(nch):
            for _ in range(CHUNK):
                step()
            evs[i + 1].record(stream)
        torch.cuda.synchronize(dev)
        per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(nch)]) * 1e3 / CHUNK  # us per launch, chunk means
        barrier()
        total = (B if scaling == "strong" else world * B) * steps
        return {"x": x, "d_out": d_out, "N": N, "elapsed": elapsed, "fastest": fastest, "gpu_ms": gpu_ms, "per_launch_us": per, "solves": total,
                "warm_launches": warm_launches}

    scaling = args.scaling
    r = run_mode(scaling, args.steps, args.warmup, args.clock_warm_ms)
    if args.dump:
        os.makedirs(args.dump, exist_ok=True)
        np.save(os.path.join(args.dump, "out_rank%d.npy" % rank), r["d_out"][:r["N"]].cpu().numpy())
    other = None
    if world > 1 and not args.no_extras:  # the other scaling mode as a second key (same protocol, clocks already warm)
        other_mode = "weak" if scaling == "strong" else "strong"
        o = run_mode(other_mode, args.steps, args.warmup, 0.0)
        other = {"scaling": other_mode, "value": o["solves"] / o["elapsed"], "unit": "solves/s", "ms_per_step": 1e3 * o["elapsed"] / args.steps,
                 "batch_per_gpu": o["N"], "global_batch": B if other_mode == "strong" else world * B}

    if rank == 0:
        N = r["N"]
        bytes_per_solve = 4 * (3 * n + 2 * n * n)   # 476 B for n = 7
        launch_ms = r["gpu_ms"] / args.steps          # average launch duration on the launch stream (HIP events), incl. launch boundaries
        achieved = bytes_per_solve * N / (launch_ms * 1e-3) / 1e9
        per = r["per_launch_us"]
        pmc, traffic, valu_insts = None, None, None
        try:
            pmc = json.load(open(os.path.join(REPO, PMC_FILE)))
            if N == pmc.get("batch"):
                traffic, valu_insts = pmc.get("hbm_bytes_per_launch"), pmc.get("sq_insts_valu_per_launch")
        except Exception:
            pmc = None
        line = {
            "metric": "forward_dynamics_gradient solves/sec, iiwa-14 batch=16384",
            "value": r["solves"] / r["elapsed"],
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * r["elapsed"] / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "iiwa14 (7-DoF chain) forward_dynamics_gradient, global batch %d, device-resident q_qd_u -> df_du" % (B if scaling == "strong" else world * B),
                       "robot": ROBOT, "batch_per_gpu": N, "global_batch": B if scaling == "strong" else world * B,
                       "sharding": "contiguous ranges of the batch axis (ceil(B/G) solves per GPU), no collective" if scaling == "strong" else "one batch per GPU, no collective"},
            "clock_warm_ms": args.clock_warm_ms, "clock_warm_launches": r["warm_launches"],
            # wall time of the K steps on the fastest and on the slowest rank (the job's time is the slowest): a straggler shows as a wide gap
            "per_rank_ms": {"min": 1e3 * r["fastest"], "max": 1e3 * r["elapsed"], "per_step_min": 1e3 * r["fastest"] / args.steps, "per_step_max": 1e3 * r["elapsed"] / args.steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": (PMC_FILE + " (rocprofv3 --pmc passes of this kernel build, not measured in this run)") if traffic else None,
                         "kernel": "forward_dynamics_gradient_kernel<float>", "launch_us": 1e3 * launch_ms,
                         "launch_us_median": float(np.median(per)), "launch_us_min": float(per.min()), "launch_us_max": float(per.max()),
                         "launch_us_note": "launch_us = event pair around the K timed launches / K; median/min/max = separate diagnostic pass, one event every 4 launches (span / 4; a marker after every launch costs ~3 us)",
                         "algorithmic_bytes_per_launch": bytes_per_solve * N,
                         "note": "the path is fp32-VALU bound (AI ~110 flop/B): see valu_issue_frac"},
            # solves/s x the REFERENCE's 53 kflop per solve / peak fp32: a reference-equivalent rate, NOT the utilisation of this kernel
            "ref_equiv_valu_frac": (N / (launch_ms * 1e-3)) * FLOPS_PER_SOLVE_REFERENCE / (FP32_PEAK_TFLOPS * 1e12),
            # VALU issue utilisation: wave-instructions (SQ_INSTS_VALU, committed PMC pass) x 2 issue cycles / (SIMDs x kernel cycles at 2.4 GHz)
            "valu_issue_frac": (valu_insts * 2.0 / (N_SIMDS * launch_ms * 1e-3 * CLOCK_GHZ * 1e9)) if valu_insts else None,
            "valu_issue_frac_source": (PMC_FILE + ": sq_insts_valu_per_launch x 2 cycles / (1024 SIMDs x launch_us x 2.4 GHz)") if valu_insts else None,
        }
        if other is not None:
            line["other_scaling"] = other
        if world == 1 and not args.no_extras:
            line["copy_bw_measured_GBps"] = measured_copy_bandwidth(torch, dev)
            line["roofline"]["frac_of_measured_copy_bw"] = achieved / line["copy_bw_measured_GBps"]
            # end-to-end incl. H2D/D2H through the host-buffer C-ABI entry point (pageable NumPy memory); never `value`
            xh = r["x"]
            lib.forward_dynamics_gradient_host(xh)
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                lib.forward_dynamics_gradient_host(xh)
            line["end_to_end_solves_per_s"] = reps * xh.shape[0] / (time.perf_counter() - t0)
            line["end_to_end_note"] = "grid_forward_dynamics_gradient_host: H2D + kernel + D2H per call, pageable host memory (fresh result array per call), PCIe-inclusive"
            # the same call on page-locked buffers (grid_host_alloc): the entry point cuts the batch into chunks and overlaps H2D | kernel | D2H on three streams
            xp, op = lib.pinned_empty(xh.shape), lib.pinned_empty((xh.shape[0], 2 * n * n))
            xp[:] = xh
            lib.forward_dynamics_gradient_host(xp, out=op)
            t0 = time.perf_counter()
            for _ in range(reps):
                lib.forward_dynamics_gradient_host(xp, out=op)
            line["end_to_end_pinned_solves_per_s"] = reps * xh.shape[0] / (time.perf_counter() - t0)
            line["end_to_end_pinned_note"] = "same call, caller's buffers from grid_host_alloc (page-locked): chunked H2D | kernel | D2H pipeline; the %.1f MB D2H at PCIe rate is the floor" % (xh.shape[0] * 2 * n * n * 4 / 1e6)
            assert np.array_equal(op, lib.forward_dynamics_gradient_host(xh))
            del xp, op
        if world == 1 and not args.no_extras:
            try:  # (diagnostics: a failure here must not cost the headline line)
                line["other_configs"] = other_configs(torch)
            except Exception as exc:
                line["other_configs"] = {"error": repr(exc)}
        if not args.no_cpu_baseline and world == 1:
            # the only place the oracle (test infrastructure) is touched: the CPU baseline leg, which also spot-checks the GPU result
            line["cpu_baseline"] = cpu_baseline(robot, r["x"])
            from oracle.rbd_oracle import Oracle

            got = r["d_out"][:32].cpu().numpy()
            ref, _ = Oracle(robot).fd_grad_batch(r["x"][:32].astype(np.float64))
            err = float(max(np.abs(got[k] - ref[k]).max() / np.abs(ref[k]).max() for k in range(32)))
            assert args.no_parity or err <= 1e-4, "parity check failed: %g" % err
            line["cpu_baseline"]["gpu_parity_max_rel_err_vs_fp64_oracle"] = err
        elif not args.no_cpu_baseline:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    lib.close()


if __name__ == "__main__":
    main()
