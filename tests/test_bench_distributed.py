"""World-size-2 gloo test (CPU) of bench.py's multi-rank harness: rendezvous on 127.0.0.1, per-rank shards, barrier,
MAX-over-ranks timing and a single JSON line from rank 0.  The data path itself has no collective (the batch axis shards)."""
import json
import os
import socket
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _selftest(world, extra):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world), "--steps", "5", "--warmup", "1", "--harness-selftest"] + extra,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines = [[l for l in o[0].splitlines() if l.startswith("{")] for o in outs]
    assert len(lines[0]) == 1 and all(len(l) == 0 for l in lines[1:])  # rank 0 prints ONE line, other ranks none
    return json.loads(lines[0][0])


def test_two_rank_harness_over_gloo_strong_scaling_is_the_default():
    """SURVEY.md section 8(e) / BASELINE.md section 2: ONE global batch of 16384 solves, rank r owns the contiguous range [r*B/G, (r+1)*B/G)."""
    d = _selftest(2, [])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["ranges"] == [[0, 8192], [8192, 16384]] and d["rows"] == [8192, 8192]
    assert d["contiguous_cover"] is True and d["shards_sum_to_global_batch"] is True
    assert d["solves_counted"] == 16384 * 5  # total work is fixed: the global batch, not world x batch
    assert d["max_elapsed_s"] >= 0.02  # the slower rank (sleeps 20 ms) sets the time


def test_three_rank_harness_ragged_split():
    d = _selftest(3, ["--batch", "1000"])
    assert d["ranges"] == [[0, 334], [334, 668], [668, 1000]] and d["rows"] == [334, 334, 332]
    assert d["contiguous_cover"] is True and d["shards_sum_to_global_batch"] is True and d["solves_counted"] == 1000 * 5


def test_eight_rank_harness_is_the_drivers_split():
    """What the driver's N = 8 run does with the batch (16 384 -> 2 048 per GPU): eight contiguous ranges, one JSON line, the
    slowest rank's time, and the per-rank spread (min / max over ranks) that makes a straggler visible in the first real scaling curve."""
    d = _selftest(8, [])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["ranges"] == [[2048 * r, 2048 * (r + 1)] for r in range(8)] and d["rows"] == [2048] * 8
    assert d["contiguous_cover"] is True and d["shards_sum_to_global_batch"] is True and d["solves_counted"] == 16384 * 5
    assert d["max_elapsed_s"] >= 0.08  # rank 7 sleeps 80 ms
    lo, hi = d["per_rank_ms"]["min"], d["per_rank_ms"]["max"]
    assert 8.0 <= lo < 40.0 and hi >= 80.0 and abs(hi - 1e3 * d["max_elapsed_s"]) < 1.0  # rank 0 slept 10 ms, rank 7 80 ms


def test_two_rank_harness_over_gloo_weak_scaling():
    d = _selftest(2, ["--scaling", "weak"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["distinct_shards"] is True
    assert d["solves_counted"] == 2 * 16384 * 5


def test_bench_line_schema_is_complete():
    """Static check of the JSON keys bench.py emits (the GPU run itself is exercised by the driver / -m gpu)."""
    src = open(os.path.join(REPO, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "bound", "achieved", "peak", "frac", "traffic", "cores", "kind", "sample",
                "launch_us_median", "clock_warm_ms", "ref_equiv_valu_frac", "valu_issue_frac", "traffic_source", "copy_bw_measured_GBps",
                "end_to_end_solves_per_s", "single_thread_fp32", "single_thread_fp64", "other_scaling", "per_rank_ms"):
        assert '"%s"' % key in src, key


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_rank_bench_rehearsal_on_one_gpu(tmp_path):
    """The REAL multi-rank code path of bench.py (per-rank shard, library per rank, opening barrier, K launches, per-rank clock stop, MAX-reduce, one JSON
    line) with two ranks sharing the one GPU of the test box: GRID_BENCH_REHEARSAL=1 swaps RCCL for gloo, nothing else.  Strong scaling (the
    default): rank r processes [r*B/G, (r+1)*B/G) of ONE global input, and the concatenated outputs equal the 1-rank result bit for bit."""
    import numpy as np

    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GRID_BENCH_REHEARSAL="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "10", "--dump", str(tmp_path / "two")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["steps"] == 100 and d["scaling"] == "strong" and d["cpu_baseline"] is None
    assert d["config"]["global_batch"] == 16384 and d["config"]["batch_per_gpu"] == 8192
    assert abs(d["value"] - 16384 * 100 / (d["ms_per_step"] * 1e-3 * 100)) <= 1e-6 * d["value"]
    assert d["value"] > 1e8
    assert d["other_scaling"]["scaling"] == "weak" and d["other_scaling"]["global_batch"] == 2 * 16384 and d["other_scaling"]["value"] > 1e8
    one = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-extras",
                          "--dump", str(tmp_path / "one")], capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    whole = np.load(tmp_path / "one" / "out_rank0.npy")
    parts = np.concatenate([np.load(tmp_path / "two" / ("out_rank%d.npy" % r)) for r in range(2)])
    assert whole.shape == (16384, 98) and np.array_equal(parts, whole)


@pytest.mark.gpu
def test_bench_rccl_code_path_with_one_rank():
    """The production multi-GPU path (backend nccl = RCCL: process-group init bound to the device, GPU barrier, MAX all-reduce of a device
    tensor) with a single rank under torch.distributed.run - all a one-GPU box can exercise of it (RCCL refuses two ranks on one device)."""
    port = free_port()
    env = dict(os.environ, GRID_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "100", "--warmup", "10", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 1e8 and d["scaling"] == "strong" and d["config"]["batch_per_gpu"] == 16384
