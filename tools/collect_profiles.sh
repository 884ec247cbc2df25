#!/bin/bash
# usage (on the GPU box): tools/collect_profiles.sh <out-dir>
# Collects what profiles/ is built from: the bench line under the driver's protocol, the rocprofv3 kernel-trace summary of the same
# command, the PMC passes of the headline kernel (each counter set in its own run, --pmc never combined with API traces), and kernel
# stats + counters of the secondary configurations (ONE batch size per output directory: 30-DoF humanoid @16384, quadruped @4096,
# second order on the 7-DoF arm @65536).
O=${1:-gpurun_out/prof}; mkdir -p $O; export TMPDIR=/tmp
B="python3 bench.py --gpus 1 --steps 20 --warmup 5"
$B > $O/bench.json 2> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- $B --no-cpu-baseline --no-extras --no-parity > $O/bench_under_rocprof.json 2> $O/trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o run -- $B --no-cpu-baseline --no-extras --no-parity --clock-warm-ms 0 > /dev/null 2> $O/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o run -- $B --no-cpu-baseline --no-extras --no-parity --clock-warm-ms 0 > /dev/null 2> $O/pmc_write.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq -o run -- $B --no-cpu-baseline --no-extras --no-parity --clock-warm-ms 0 > /dev/null 2> $O/pmc_sq.err || exit 1
for cfg in "atlas 16384" "hyq 4096"; do
  set -- $cfg; R=$1; N=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_trace -o run -- python3 tools/bench_variant.py $R $N - > $O/${R}_bench.json 2> $O/${R}_trace.err || exit 1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${R}_pmc_fetch -o run -- python3 tools/bench_variant.py $R $N - > /dev/null 2> $O/${R}_pmc_fetch.err || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${R}_pmc_write -o run -- python3 tools/bench_variant.py $R $N - > /dev/null 2> $O/${R}_pmc_write.err || exit 1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/${R}_pmc_sq -o run -- python3 tools/bench_variant.py $R $N - > /dev/null 2> $O/${R}_pmc_sq.err || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/so_trace -o run -- python3 tools/bench_idsva_so.py iiwa14 65536 > $O/so_bench.jsonl 2> $O/so_trace.err || exit 1
python3 tools/bench_idsva_so.py hyq 4096 >> $O/so_bench.jsonl 2> /dev/null || exit 1
python3 tools/bench_idsva_so.py atlas 1024 >> $O/so_bench.jsonl 2> /dev/null || exit 1
python3 tests/tools/bench_components.py > $O/components.jsonl 2> /dev/null || exit 1
python3 tools/bench_robots.py > $O/robots.jsonl 2> /dev/null || exit 1
for cfg in "atlas 16384" "hyq 4096" "iiwa14 1024" "iiwa14 16384"; do python3 tools/bench_kernels.py $cfg >> $O/kernels.jsonl 2> /dev/null || exit 1; done
echo done
