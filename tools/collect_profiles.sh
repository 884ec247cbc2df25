#!/bin/bash
# usage (on the GPU box): tools/collect_profiles.sh <out-dir>
# Collects what profiles/ is built from: the bench line, the rocprofv3 kernel-trace summary of the same command, and the PMC passes
# (each counter set in its own run, --pmc never combined with API traces).
O=${1:-gpurun_out/prof}; mkdir -p $O; export TMPDIR=/tmp
python3 bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $O/bench_under_rocprof.json 2> $O/trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > /dev/null 2> $O/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > /dev/null 2> $O/pmc_write.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > /dev/null 2> $O/pmc_sq.err || exit 1
python3 tests/tools/bench_components.py > $O/components.jsonl 2> /dev/null || exit 1
echo done
