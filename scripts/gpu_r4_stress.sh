#!/bin/bash
# stress configuration (BASELINE configs[4]): bench line + kernel trace
tag=${1:-r4}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --config stress --no-cpu-baseline --no-dropin --no-extra --windows 3 > $out/${tag}_bench_stress.json 2> $out/${tag}_bench_stress.err
python -c "
import json;d=json.load(open('$out/${tag}_bench_stress.json'));print('stress', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_prof_stress -- python3 $GRAFT_REPO_ROOT/bench.py --config stress --steps 60 --warmup 10 --no-cpu-baseline --no-dropin --no-extra --windows 1 > $GRAFT_REPO_ROOT/$out/${tag}_prof_stress.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT && python scripts/kstats.py $out/${tag}_prof_stress 10 > $out/${tag}_kernel_stats_stress.txt 2>&1; cat $out/${tag}_kernel_stats_stress.txt
