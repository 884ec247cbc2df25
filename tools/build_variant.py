#!/usr/bin/env python3
"""Builds an experimental variant of a robot library into gridcodegenerator_amd/_build_x/<tag>/ (git-ignored, travels with gpurun).
usage: python tools/build_variant.py <robot> <tag> [key=value ...] [-- extra hipcc flags]
Keys are GRiDCodeGenerator tuning keys (TUNING_DEFAULTS / TUNING_ABLATION); ablation keys need allow_wrong_results=1.
The timing tools take the variant's directory: tools/bench_variant.py <robot> <N> gridcodegenerator_amd/_build_x/<tag>"""
import ast
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gridcodegenerator_amd.runtime import PKG_DIR, build_library  # noqa: E402


def parse(v):
    try:
        return ast.literal_eval(v)
    except Exception:
        return v


def main():
    args = sys.argv[1:]
    flags = []
    if "--" in args:
        i = args.index("--")
        args, flags = args[:i], args[i + 1:]
    robot, tag = args[0], args[1]
    tuning = {k: parse(v) for k, v in (a.split("=", 1) for a in args[2:])}
    out = os.path.join(PKG_DIR, "_build_x", tag)
    so = build_library(robot, build_dir=out, tuning=tuning, extra_flags=flags, force=True)
    print(so)


if __name__ == "__main__":
    main()
