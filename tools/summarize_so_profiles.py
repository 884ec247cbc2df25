#!/usr/bin/env python3
"""Turns one tools/collect_so_profiles.sh output directory into the files kept under profiles/:  <tag>_so_<robot>_kernel_stats.csv (rocprofv3 --stats),
<tag>_so_<robot>_pmc.txt (per-launch counters of idsva_so_kernel and fdsva_so_kernel with the HBM traffic corrected as MI355X_MICROARCH.md prescribes)
and <tag>_so_<robot>_bench.jsonl.   usage: tools/summarize_so_profiles.py <collect-dir> <tag> [robot batch]..."""
import collections, csv, glob, json, os, shutil, sys

base, tag = sys.argv[1].rstrip("/") + "/", sys.argv[2]
cfgs = sys.argv[3:] or ["iiwa14", "65536", "atlas", "1024"]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
KERNELS = ("idsva_so_kernel", "fdsva_so_kernel", "fdsva_so_prepare_kernel", "fdsva_so_contract_kernel")  # (the humanoid's fdsva_so runs as prepare + contract)
N_OF = {"iiwa14": 7, "arm6": 6, "hyq": 12, "tree12": 12, "atlas": 30, "chain8": 8, "chain12": 12, "mixed5": 5}


def counters(robot, kernel_substr):
    out = {}
    for part in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        fs = glob.glob(base + robot + "_" + part + "/**/*counter_collection.csv", recursive=True)
        if not fs:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if kernel_substr in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = sum(v) / len(v)
    return out


for robot, N in zip(cfgs[0::2], (int(x) for x in cfgs[1::2])):
    fs = glob.glob(base + robot + "_trace/**/*kernel_stats.csv", recursive=True)
    if not fs:
        continue
    shutil.copy(fs[0], os.path.join(P, "%s_so_%s_kernel_stats.csv" % (tag, robot)))
    shutil.copy(base + robot + "_bench.jsonl", os.path.join(P, "%s_so_%s_bench.jsonl" % (tag, robot)))
    n = N_OF[robot]
    alg = 4 * (3 * n + 4 * n ** 3) * N
    dur = {}
    for r in csv.DictReader(open(fs[0])):
        for k in KERNELS:
            if k + "<" in r["Name"]:
                dur[k] = float(r["AverageNs"])
    with open(os.path.join(P, "%s_so_%s_pmc.txt" % (tag, robot)), "w") as f:
        f.write("rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/bench_idsva_so.py %s %d   (tools/collect_so_profiles.sh; per launch of %d solves; separate passes)\n" % (robot, N, N))
        f.write("algorithmic HBM bytes per launch: 4*(3n + 4n^3)*N = %d\n" % alg)
        for k in KERNELS:
            out = counters(robot, k + "<")
            if not out:
                continue
            w = out.get("SQ_WAVE_CYCLES", 0.0)
            f.write("\n== %s: rocprofv3 --stats average %.1f us -> %.1f GB/s algorithmic = %.3f of 8 TB/s\n" % (k, dur.get(k, 0) / 1e3, alg / max(dur.get(k, 1), 1), alg / max(dur.get(k, 1), 1) / 8000))
            for c, v in sorted(out.items()):
                if c.startswith("SQ_"):
                    f.write("%-26s %16.0f   %5.1f %% of SQ_WAVE_CYCLES\n" % (c, v, 100 * v / w if w else 0))
            if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
                hbm = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024
                f.write("FETCH_SIZE (KB, raw) %12.1f\nWRITE_SIZE (KB, raw) %12.1f\nHBM bytes per launch (2 x FETCH + WRITE, MI355X_MICROARCH.md correction) %.0f = %.2f x algorithmic\n"
                        % (out["FETCH_SIZE"], out["WRITE_SIZE"], hbm, hbm / alg))
    print(open(os.path.join(P, "%s_so_%s_pmc.txt" % (tag, robot))).read())
    print(open(fs[0]).read())
