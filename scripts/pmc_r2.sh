#!/bin/bash
# PMC passes over the update workload (scripts/profile_traffic.py), one rocprofv3 run per pass with --kernel-trace only;
# every pass keeps its log, a failing pass does not stop the others.
# usage: bash scripts/pmc_r2.sh <outdir-under-gpurun_out> [config] [pass letters, default all]
out=gpurun_out/$1
cfg=${2:-default}
which=${3:-abcdefghij}
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
(rocprofv3 -L > $R/$out/counters_available.txt 2>&1) || true
run() {
  case $which in *$1*) ;; *) return;; esac
  timeout -k 10 240 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/$out/$1 -o $1 -- python3 $R/scripts/profile_traffic.py $cfg > $R/$out/$1.log 2>&1
  echo "pass $1 ($2) rc=$?"
}
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU"
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH"
run d "FETCH_SIZE"
run e "WRITE_SIZE"
run f "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
run g "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
run h "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
run i "TA_BUSY_avr TA_BUFFER_READ_WAVEFRONTS_sum"
run j "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
cd $R
for p in a b c d e f g h i j; do [ -d $out/$p ] && python3 scripts/pmc_summary.py $out/$p corr_fused; done > $out/summary.txt
cat $out/summary.txt
