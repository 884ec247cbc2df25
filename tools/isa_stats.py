#!/usr/bin/env python3
"""Static instruction statistics of a robot library's kernels (no GPU needed): compiles the generated header for gfx950 to assembly and counts
VALU / LDS / SALU / VMEM instructions, VGPRs and scratch per kernel.  usage: python tools/isa_stats.py <robot> [key=value tuning ...] [--kernel substr]"""
import ast
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gridcodegenerator_amd import RobotModel  # noqa: E402
from gridcodegenerator_amd.runtime import CAPI_SRC, HIPCC_FLAGS, INCLUDE_DIR, generate_header  # noqa: E402


def asm_for(robot, tuning=None, keep=None):
    d = keep or tempfile.mkdtemp(prefix="grid_isa_")
    generate_header(RobotModel.from_fixture(robot) if isinstance(robot, str) else robot, d, tuning=tuning)
    out = os.path.join(d, "grid.s")
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["--offload-device-only", "-S", "-I" + d, "-I" + INCLUDE_DIR, '-DGRID_ROBOT_NAME="x"', CAPI_SRC, "-o", out])
    return out


def stats(asm_path, only=None):
    txt = open(asm_path).read()
    res = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\n\s*s_endpgm", txt, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        if only and only not in name:
            continue
        c = {"valu": 0, "lds": 0, "salu": 0, "vmem": 0, "dpp": 0, "trans": 0, "waitcnt": 0, "nop": 0, "f64": 0}
        for line in body.split("\n"):
            t = line.strip().split()
            if not t or t[0].startswith(";") or t[0].endswith(":") or t[0].startswith("."):
                continue
            op = t[0]
            if op.startswith("v_"):
                c["valu"] += 1
                if "dpp" in line:
                    c["dpp"] += 1
                if re.match(r"v_(rcp|sin|cos|sqrt|rsq|exp|log)", op):
                    c["trans"] += 1
                if "_f64" in op:
                    c["f64"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith("s_waitcnt"):
                c["waitcnt"] += 1
            elif op.startswith("s_nop"):
                c["nop"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
        meta = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(name), txt, flags=re.S)
        if meta:
            for key in ("next_free_vgpr", "private_segment_fixed_size", "accum_offset"):
                mm = re.search(r"\.amdhsa_%s (\d+)" % key, meta.group(1))
                if mm:
                    c[key] = int(mm.group(1))
        res[name] = c
    return res


if __name__ == "__main__":
    args = sys.argv[1:]
    only = None
    if "--kernel" in args:
        i = args.index("--kernel")
        only = args[i + 1]
        args = args[:i] + args[i + 2:]
    robot = args[0]
    tuning = {}
    for a in args[1:]:
        k, v = a.split("=", 1)
        try:
            v = ast.literal_eval(v)
        except Exception:
            pass
        tuning[k] = v
    path = asm_for(robot, tuning or None)
    for name, c in stats(path, only).items():
        filt = next((f for f in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "/usr/bin/c++filt") if os.path.exists(f)), None)
        dem = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else name
        print(dem[:110])
        print("   ", c)
    print(path)
