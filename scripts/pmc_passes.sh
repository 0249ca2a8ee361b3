#!/bin/bash
# PMC passes over the update workload (scripts/profile_traffic.py); each pass is its own rocprofv3 run with
# --kernel-trace only.  usage: bash scripts/pmc_passes.sh <outdir-under-gpurun_out> ; summaries: scripts/pmc_summary.py
set -e
out=gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/$out/$1 -o $1 -- python3 $R/scripts/profile_traffic.py > $R/$out/$1.log 2>&1; }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU"
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
# (a pass over TA_* / TCP_* / TD_* sums aborted inside the profiler on this pool: not collected)
cd $R
for p in a b c; do python3 scripts/pmc_summary.py $out/$p corr_fused; done > $out/summary.txt
cat $out/summary.txt
