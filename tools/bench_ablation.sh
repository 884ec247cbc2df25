#!/bin/bash
# usage: tools/bench_ablation.sh <robot> <batch> <build-dir>...   (timing of experimental builds; "-" = the shipped build)
R=$1; N=$2; shift 2
for d in "$@"; do python tools/bench_variant.py $R $N $d || exit 1; done
