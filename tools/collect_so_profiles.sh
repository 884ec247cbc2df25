#!/bin/bash
# usage (on the GPU box): tools/collect_so_profiles.sh <out-dir> [robot batch]...
# Second-order kernels (idsva_so / fdsva_so): rocprofv3 kernel-trace summary plus the PMC passes (each counter set in its own run, --pmc never
# combined with API traces) of `tools/bench_idsva_so.py <robot> <batch>`, ONE robot and batch size per output sub-directory.
O=${1:-gpurun_out/so_prof}; shift; mkdir -p $O; export TMPDIR=/tmp
[ $# -eq 0 ] && set -- iiwa14 65536 atlas 1024
while [ $# -ge 2 ]; do
  R=$1; N=$2; shift 2
  B="python3 tools/bench_idsva_so.py $R $N"
  $B > $O/${R}_bench.jsonl 2> $O/${R}_bench.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_trace -o run -- $B > /dev/null 2> $O/${R}_trace.err || exit 1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${R}_pmc_fetch -o run -- $B > /dev/null 2> $O/${R}_pmc_fetch.err || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${R}_pmc_write -o run -- $B > /dev/null 2> $O/${R}_pmc_write.err || exit 1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/${R}_pmc_sq -o run -- $B > /dev/null 2> $O/${R}_pmc_sq.err || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/${R}_pmc_sq2 -o run -- $B > /dev/null 2> $O/${R}_pmc_sq2.err || echo "(second SQ counter set not available)" > $O/${R}_pmc_sq2.note
  echo "$R $N done"
done
echo done
