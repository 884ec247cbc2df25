#!/bin/bash
# usage: tools/prof_ablation.sh <robot> <batch> <out-dir> <build-dir>...   rocprofv3 kernel durations of experimental builds ("-" = shipped build)
R=$1; N=$2; O=$3; shift 3
mkdir -p $O; export TMPDIR=/tmp
for d in "$@"; do
  tag=$(basename $d); [ "$d" = "-" ] && tag=default
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o run -- python3 tools/bench_variant.py $R $N $d > $O/$tag.log 2>&1 || exit 1
  f=$(find $O/$tag -name "*kernel_stats.csv" | sort | sed -n 1p)
  if [ -z "$f" ]; then echo "$tag N=$N: no kernel_stats.csv"; exit 1; fi
  echo "$tag N=$N: $(grep forward_dynamics_gradient "$f" < /dev/null | cut -d, -f2-4,6-7 | tr -d '"')"
done
