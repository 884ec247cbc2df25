#!/bin/bash
# usage: tools/prof_threads.sh <robot> <batch> <out-dir> <build-dir> <threads>...   rocprofv3 kernel durations of one build at several block sizes
R=$1; N=$2; O=$3; D=$4; shift 4
mkdir -p $O; export TMPDIR=/tmp
for t in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$t -o run -- python3 tools/bench_variant.py $R $N $D $t > $O/t$t.log 2>&1 || exit 1
  f=$(find $O/t$t -name "*kernel_stats.csv" | sort | sed -n 1p)
  python3 - "$f" "threads=$t" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "forward_dynamics_gradient" in r["Name"]:
        print("%-14s calls %s avg %.2f us  min %.2f  max %.2f" % (sys.argv[2], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
