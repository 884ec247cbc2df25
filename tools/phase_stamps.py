#!/usr/bin/env python3
"""Reads the per-wave cycle stamps a tuning debug_stop=20 build (tools/build_variant.py iiwa14 stamps debug_stop=20 allow_wrong_results=1) of the tip-frame kernel leaves in the first outputs of every solve.
usage: python tools/phase_stamps.py <build-dir> [batch] [robot]   (robot other than iiwa14: the branch-frame kernel's 8 stamps)"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from gridcodegenerator_amd.runtime import load
bdir = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
robot = sys.argv[3] if len(sys.argv) > 3 else "iiwa14"
lib = load(robot, max_timesteps=N, build_dir=bdir)
rng = np.random.default_rng(0); n = lib.n
x = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((N, 2 * n * n), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(20): lib.forward_dynamics_gradient_device(d_in, N, d_out, stream=st)
torch.cuda.synchronize()
if robot != "iiwa14":
    o = d_out.cpu().numpy()[::64 // lib.lanes_per_solve]
    names = ["table row, zero fill", "frame chain + sync", "link set-up, v, a (walk)", "body, scans, branch hand-over", "pass 1 (M, dc/dqd) + sync", "factorisation, qdd + syncs", "pass 2 (da, f^C, dc/dq) + sync", "column solves"]
    ph = np.diff(o[:, 0:9], axis=1)
    print("%s: waves %d; cycles per phase (mean / min / max over waves), total inner %.0f cycles mean" % (robot, o.shape[0], o[:, 8].mean()))
    for i, nm in enumerate(names):
        print("  %-45s %7.0f %7.0f %7.0f" % (nm, ph[:, i].mean(), ph[:, i].min(), ph[:, i].max()))
    sys.exit(0)
o = d_out.cpu().numpy()[::8]  # one row per wave (8 solves of a wave share their stamps)
names = ["chain (+link constants)", "link set-up, bias, scans, record + sync", "pass 1, M write + sync", "M read, factorisation, qdd", "qdd-dependent part, pass 2, two solves", "output staging + sync"]
ph = np.diff(o[:, 0:7], axis=1)
print("waves %d; cycles per phase (mean / min / max over waves), total inner %.0f cycles mean" % (o.shape[0], o[:, 6].mean()))
for i, nm in enumerate(names):
    print("  %-45s %7.0f %7.0f %7.0f" % (nm, ph[:, i].mean(), ph[:, i].min(), ph[:, i].max()))
entry = o[:, 7]; e0 = entry.min()
rel = np.mod(entry - e0, 2 ** 24)
print("inner-entry stamp relative to the earliest wave: mean %.0f  p50 %.0f  p90 %.0f  max %.0f cycles" % (rel.mean(), np.median(rel), np.quantile(rel, 0.9), rel.max()))
end = rel + o[:, 6]
print("inner-exit  stamp relative to the earliest entry: mean %.0f  p50 %.0f  p90 %.0f  max %.0f cycles" % (end.mean(), np.median(end), np.quantile(end, 0.9), end.max()))
