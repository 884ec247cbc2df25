import sys, time, os
sys.path.insert(0, ".")  # run from the repo root
import numpy as np
import bench
from gridcodegenerator_amd import RobotModel
from oracle import rbd_oracle
robot = RobotModel.from_fixture("iiwa14")
x = bench.make_inputs(7, 16384)
so = rbd_oracle.build(march="native", out="/tmp/librbd_native.so", force=True)
orc = rbd_oracle.Oracle(robot, dtype=np.float32, lib_path=so)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for nt in (1, 4, 8, 16, 32, 64, 128):
    orc.fd_grad_batch(x, nthreads=nt)
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 2.0:
        orc.fd_grad_batch(x, nthreads=nt); reps += 1
    dt = time.perf_counter() - t0
    print(nt, "threads:", round(reps * 16384 / dt), "solves/s")
