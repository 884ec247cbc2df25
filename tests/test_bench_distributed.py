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


def test_two_rank_harness_over_gloo():
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--harness-selftest"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    lines1 = [l for l in outs[1][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and len(lines1) == 0  # rank 0 prints ONE line, other ranks none
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["distinct_shards"] is True
    assert d["solves_counted"] == 2 * 16384 * 5
    assert d["max_elapsed_s"] >= 0.02  # the slower rank (sleeps 20 ms) sets the time


def test_bench_line_schema_is_complete():
    """Static check of the JSON keys bench.py emits (the GPU run itself is exercised by the driver / -m gpu)."""
    src = open(os.path.join(REPO, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "bound", "achieved", "peak", "frac", "traffic", "cores", "kind", "sample"):
        assert '"%s"' % key in src, key


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_rank_bench_rehearsal_on_one_gpu():
    """The REAL multi-rank code path of bench.py (per-rank shard, library per rank, opening barrier, K launches, per-rank clock stop, MAX-reduce, one JSON
    line) with two ranks sharing the one GPU of the test box: GRID_BENCH_REHEARSAL=1 swaps RCCL for gloo, nothing else."""
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GRID_BENCH_REHEARSAL="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "10"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["steps"] == 100 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["global_batch"] == 2 * 16384
    assert abs(d["value"] - 2 * 16384 * 100 / (d["ms_per_step"] * 1e-3 * 100)) <= 1e-6 * d["value"]
    assert d["value"] > 1e8


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
    assert d["n_gpus"] == 1 and d["value"] > 1e8 and d["scaling"] == "weak"
