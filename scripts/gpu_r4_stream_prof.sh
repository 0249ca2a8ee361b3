#!/bin/bash
# kernel trace of the device-resident stream
tag=${1:-r4}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python scripts/bench_stream.py 600 > $out/${tag}_stream.log 2>&1; tail -2 $out/${tag}_stream.log
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_stream_prof -- python3 $GRAFT_REPO_ROOT/scripts/bench_stream.py 300 > $GRAFT_REPO_ROOT/$out/${tag}_stream_prof.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT && python scripts/kstats.py $out/${tag}_stream_prof 30 > $out/${tag}_kernel_stats_stream.txt 2>&1; cat $out/${tag}_kernel_stats_stream.txt
