#!/usr/bin/env python3
"""Where does the fp32 error of the tip-frame forward-dynamics gradient come from?  (CPU only: runs the generated code in the test emulation.)

The T = double instantiation of the generated kernel is exact to 1e-14; with tuning round_probe=(stage,) the results of ONE stage are rounded
to fp32 inside it.  The error against the fp64 oracle then is that stage's representation error propagated through the rest of the algorithm.
usage: python tools/precision_probe.py [robot] [N]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from emu_harness import emu_library  # noqa: E402
from gridcodegenerator_amd import RobotModel  # noqa: E402
from oracle.rbd_oracle import Oracle  # noqa: E402

WORST_IIWA14 = [-1.9726811647415161, -0.15061229467391968, 3.0075197219848633, -1.6245150566101074, 0.2933456301689148, -0.012131131254136562,
                -0.5525374412536621, -1.6120431423187256, -1.4713057279586792, -1.4227734804153442, 1.8603541851043701, 0.7642495632171631,
                0.44305285811424255, -1.240814447402954, -6.242157936096191, -5.802548885345459, 8.430603981018066, -4.893815517425537,
                7.717841625213623, -4.148682594299316, -4.838983535766602]
STAGES = ["chain", "link", "vel", "comp_I", "comp_BF", "t1", "t24", "rhs", "M", "pass1", "Mread", "factor", "qdd", "t3", "pass2"]


def sample(n, N, seed=5):
    rng = np.random.default_rng(seed)
    xs = np.hstack([rng.uniform(-np.pi, np.pi, (N, n)), rng.uniform(-2, 2, (N, n)), rng.uniform(-10, 10, (N, n))]).astype(np.float32)
    xs[:N // 2, 1:n:2] = rng.uniform(-0.05, 0.05, (N // 2, len(range(1, n, 2)))).astype(np.float32)
    return xs


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "iiwa14"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    robot = RobotModel.from_fixture(name)
    n = robot.n
    xs = sample(n, N)
    if name == "iiwa14":
        xs = np.vstack([np.array(WORST_IIWA14, np.float32)[None], xs])
    ref, _ = Oracle(robot).fd_grad_batch(xs.astype(np.float64))
    err = lambda out: np.abs(out - ref).max(axis=1) / np.abs(ref).max(axis=1)
    lib = emu_library(name, max_timesteps=N + 8)
    lib.set_launch_dims(0, 64)
    e = err(lib.forward_dynamics_gradient_host(xs))
    print("%-28s first %.2e  max %.2e  p99 %.2e  median %.2e" % ("fp32 kernel", e[0], e.max(), np.quantile(e, 0.99), np.median(e)))
    for st in [()] + [(s,) for s in STAGES]:
        lib = emu_library(name, max_timesteps=N + 8, tuning={"round_probe": st, "allow_wrong_results": True})
        lib.set_launch_dims(0, 64)
        e = err(lib.forward_dynamics_gradient_host_f64(xs.astype(np.float64)))
        print("%-28s first %.2e  max %.2e  p99 %.2e  median %.2e" % ("fp64 + fp32 round of %s" % (st[0] if st else "-"), e[0], e.max(), np.quantile(e, 0.99), np.median(e)))


if __name__ == "__main__":
    main()
