#!/bin/bash
# the global bundle adjustment at the reference's scale: timing + kernel trace.  usage: scripts/gpu_r3_global.sh TAG
tag=${1:-r3gb}; out=gpurun_out; export TMPDIR=/tmp; mkdir -p $out
timeout -k 10 300 python scripts/bench_global_ba.py 300 96 > $out/${tag}_global_time.log 2>&1
grep -v amdgpu $out/${tag}_global_time.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_global -- python3 $GRAFT_REPO_ROOT/scripts/bench_global_ba.py 300 96 > $GRAFT_REPO_ROOT/$out/${tag}_global.log 2>&1)
python scripts/kstats.py $out/${tag}_global 16 > $out/${tag}_kernel_stats_global.txt 2>&1; cat $out/${tag}_kernel_stats_global.txt
