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
run d "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
run e "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
cd $R
for p in a b c d e; do python3 scripts/pmc_summary.py $out/$p corr_fused; done > $out/summary.txt
cat $out/summary.txt
