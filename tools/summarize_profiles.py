#!/usr/bin/env python3
"""Turns one tools/collect_profiles.sh output directory into the files kept under profiles/ (kernel stats CSV, bench line,
PMC summary text, pmc_traffic.json).  usage: tools/summarize_profiles.py <collect-dir> <tag>   e.g.  gpurun_out/r01e r01"""
import collections, csv, glob, json, os, shutil, sys
base, tag = sys.argv[1].rstrip("/") + "/", sys.argv[2]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
ks = glob.glob(base + "trace/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(ks, os.path.join(P, tag + "_kernel_stats.csv"))
shutil.copy(base + "bench.json", os.path.join(P, tag + "_bench_n1.json"))
if os.path.exists(base + "components.jsonl"):
    shutil.copy(base + "components.jsonl", os.path.join(P, tag + "_components.jsonl"))
bench = json.load(open(base + "bench.json"))
out = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = glob.glob(base + sub + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "forward_dynamics_gradient" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
fetch_kb, write_kb = out["FETCH_SIZE"], out["WRITE_SIZE"]
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
json.dump({"hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024, "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
           "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reports 1/2 of streamed read bytes on gfx950 (x2 applied; dword-wide reads are uncalibrated), WRITE_SIZE exact for 16-byte stores; units KB",
           "algorithmic_bytes_per_launch": alg,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 20 (tools/collect_profiles.sh)"},
          open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
w = out["SQ_WAVE_CYCLES"]
with open(os.path.join(P, tag + "_pmc_sq.txt"), "w") as f:
    f.write("rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 20 --warmup 5   (tools/collect_profiles.sh; per launch of 16384 solves = 2048 waves)\n")
    for k, v in sorted(out.items()):
        if k.startswith("SQ_"):
            f.write("%-22s %12.0f   %5.1f %% of SQ_WAVE_CYCLES\n" % (k, v, 100 * v / w))
    f.write("FETCH_SIZE (KB, raw) %10.1f\nWRITE_SIZE (KB, raw) %10.1f\n" % (fetch_kb, write_kb))
    hist = os.path.join(P, "pmc_history.txt")
    if os.path.exists(hist):
        f.write("\n" + open(hist).read())
print(open(ks).read())
print(open(os.path.join(P, tag + "_pmc_sq.txt")).read())
print(json.dumps({k: bench[k] for k in ("value", "ms_per_step", "roofline", "valu_frac", "cpu_baseline")})[:900])
