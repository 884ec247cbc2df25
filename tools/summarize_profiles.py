#!/usr/bin/env python3
"""Turns one tools/collect_profiles.sh output directory into the files kept under profiles/ (kernel stats CSVs with ONE batch size each,
the bench line, PMC summaries, pmc_counters.json that bench.py reads).  usage: tools/summarize_profiles.py <collect-dir> <tag>   e.g.  gpurun_out/r02p r02"""
import collections, csv, glob, json, os, shutil, sys
base, tag = sys.argv[1].rstrip("/") + "/", sys.argv[2]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def counters(sub, kernel_substr):
    out = {}
    for part in ("pmc_fetch", "pmc_write", "pmc_sq"):
        fs = glob.glob(base + sub + part + "/**/*counter_collection.csv", recursive=True)
        if not fs:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if kernel_substr in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = sum(v) / len(v)
    return out


def write_sq(path, head, out):
    w = out.get("SQ_WAVE_CYCLES", 0.0)
    with open(path, "w") as f:
        f.write(head + "\n")
        for k, v in sorted(out.items()):
            if k.startswith("SQ_"):
                f.write("%-22s %14.0f   %5.1f %% of SQ_WAVE_CYCLES\n" % (k, v, 100 * v / w if w else 0))
        if "FETCH_SIZE" in out:
            f.write("FETCH_SIZE (KB, raw) %10.1f\nWRITE_SIZE (KB, raw) %10.1f\nHBM bytes per launch (2 x FETCH + WRITE, MI355X_MICROARCH.md correction) %.0f\n"
                    % (out["FETCH_SIZE"], out["WRITE_SIZE"], (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024))


ks = glob.glob(base + "trace/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(ks, os.path.join(P, tag + "_kernel_stats.csv"))
shutil.copy(base + "bench.json", os.path.join(P, tag + "_bench_n1.json"))
for extra in ("components.jsonl", "robots.jsonl", "so_bench.jsonl", "kernels.jsonl"):
    if os.path.exists(base + extra):
        shutil.copy(base + extra, os.path.join(P, tag + "_" + extra))
bench = json.loads(open(base + "bench.json").read().strip().splitlines()[-1])
out = counters("", "forward_dynamics_gradient")
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
json.dump({"kernel": "forward_dynamics_gradient_kernel<float>", "robot": "iiwa14", "batch": 16384,
           "hbm_bytes_per_launch": (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024, "FETCH_SIZE_KB_raw": out["FETCH_SIZE"], "WRITE_SIZE_KB_raw": out["WRITE_SIZE"],
           "sq_insts_valu_per_launch": out["SQ_INSTS_VALU"], "sq_wave_cycles_per_launch": out["SQ_WAVE_CYCLES"],
           "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reports 1/2 of streamed read bytes on gfx950 (x2 applied; dword-wide reads are uncalibrated), WRITE_SIZE exact for 16-byte stores; units KB",
           "algorithmic_bytes_per_launch": alg,
           "source": "%s build: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes) around `python3 bench.py --gpus 1 --steps 20 --warmup 5` (tools/collect_profiles.sh); profiles/%s_pmc_sq.txt" % (tag, tag)},
          open(os.path.join(P, "pmc_counters.json"), "w"), indent=1)
write_sq(os.path.join(P, tag + "_pmc_sq.txt"), "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5   (tools/collect_profiles.sh; per launch of 16384 iiwa14 solves = 2048 waves)", out)
for robot in ("atlas", "hyq"):
    fs = glob.glob(base + robot + "_trace/**/*kernel_stats.csv", recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, robot)))
        shutil.copy(base + robot + "_bench.json", os.path.join(P, "%s_%s_bench.json" % (tag, robot)))
        write_sq(os.path.join(P, "%s_%s_pmc.txt" % (tag, robot)), "rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/bench_variant.py %s <N> -   (one batch size per file; per launch)" % robot,
                 counters(robot + "_", "forward_dynamics_gradient"))
fs = glob.glob(base + "so_trace/**/*kernel_stats.csv", recursive=True)
if fs:
    shutil.copy(fs[0], os.path.join(P, tag + "_second_order_kernel_stats.csv"))
print(open(ks).read())
print(open(os.path.join(P, tag + "_pmc_sq.txt")).read())
print(json.dumps({k: bench.get(k) for k in ("value", "ms_per_step", "roofline", "valu_issue_frac", "ref_equiv_valu_frac", "cpu_baseline")})[:1500])
