"""Builds the CPU *emulation* of a robot library: the unchanged generated header + the unchanged C-ABI shim compiled by
g++ against tests/emu/hip/hip_runtime.h (threads-per-lane emulation).  Test infrastructure only."""
import os
import subprocess
import tempfile

from gridcodegenerator_amd import RobotModel
from gridcodegenerator_amd.runtime import CAPI_SRC, INCLUDE_DIR, GridLibrary, generate_header

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_INC = os.path.join(HERE, "emu")
MANIFEST = os.path.join(EMU_INC, "prebuild_manifest.json")
_CACHE = {}
_REQUESTED = {}  # key -> manifest entry of every library this process asked for (pytest --update-emu-manifest writes them out)


def _key(robot, cols_per_lane, tuning, debug_mode):
    return robot.name + ("" if cols_per_lane is None else "_c%d" % cols_per_lane) + ("_debug" if debug_mode else "") + "".join("_%s%s" % (k, v) for k, v in sorted((tuning or {}).items()))


def _build(robot, key, cols_per_lane, tuning, debug_mode):
    out_dir = os.path.join(tempfile.gettempdir(), "grid_emu_build", key)
    generate_header(robot, out_dir, cols_per_lane=cols_per_lane, tuning=tuning, debug_mode=debug_mode)
    so = os.path.join(out_dir, "libgrid_emu_%s.so" % key)
    cmd = ["g++", "-std=c++20", "-O0", "-g0", "-fno-gnu-unique",  # (no process-wide "unique" symbols: inline variables and function-local statics stay private to each robot library)
           "-x", "c++", "-shared", "-fPIC", "-pthread", "-I" + EMU_INC, "-I" + out_dir, "-I" + INCLUDE_DIR,
           '-DGRID_ROBOT_NAME="%s"' % robot.name, "-Wno-unused-value", CAPI_SRC, "-o", so]
    subprocess.check_call(cmd)
    return so


def emu_library(robot, max_timesteps=64, cols_per_lane=None, tuning=None, debug_mode=False):
    """tuning: generation-time knobs (GRiDCodeGenerator(..., tuning=...)) for this build only."""
    fixture = robot if isinstance(robot, str) else None
    if isinstance(robot, str):
        robot = RobotModel.from_fixture(robot)
    key = _key(robot, cols_per_lane, tuning, debug_mode)
    _REQUESTED.setdefault(key, {"robot": fixture if fixture is not None else robot.desc, "cols_per_lane": cols_per_lane, "tuning": tuning, "debug_mode": debug_mode})
    if key not in _CACHE:
        _CACHE[key] = _build(robot, key, cols_per_lane, tuning, debug_mode)
    return GridLibrary(_CACHE[key], device=0, max_timesteps=max_timesteps)


def prebuild_from_manifest(workers):
    """Compiles the emulation libraries a full run of the CPU suite asks for (tests/emu/prebuild_manifest.json, written by `pytest --update-emu-manifest`) on
    several cores at once - g++ needs 4-6 s per library and the suite ~60 of them.  A library that is missing from the manifest is simply built when a test asks for it."""
    import json
    import multiprocessing as mp

    if not os.path.exists(MANIFEST):
        return 0
    entries = [e for e in json.load(open(MANIFEST))]
    # (worker PROCESSES: the generator writes its header through module-level state, two generations in one process would mix)
    with mp.get_context("fork").Pool(workers) as pool:
        done = [r for r in pool.map(_prebuild_one, entries, chunksize=1) if r is not None]
    _CACHE.update(dict(done))
    return len(done)


def _prebuild_one(e):
    robot = RobotModel.from_fixture(e["robot"]) if isinstance(e["robot"], str) else RobotModel(e["robot"])
    key = _key(robot, e["cols_per_lane"], e["tuning"], e["debug_mode"])
    try:
        return key, _build(robot, key, e["cols_per_lane"], e["tuning"], e["debug_mode"])
    except Exception:  # (a stale entry: the test that wants this library builds it itself and reports the error)
        return None


def write_manifest():
    import json

    with open(MANIFEST, "w") as f:
        json.dump([_REQUESTED[k] for k in sorted(_REQUESTED)], f, indent=0, sort_keys=True)
        f.write("\n")
