#!/bin/bash
# one GPU-box call: the GPU test suite, then the bench line + kernel trace for the default and the stress workload
# usage: scripts/gpu_r2_full.sh TAG
tag=${1:-r2}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
rm -f $out/${tag}_ba_errors.jsonl
export CDV_TEST_LOG=$PWD/$out/${tag}_ba_errors.jsonl
timeout -k 10 1000 python -m pytest tests -m gpu -v --timeout 600 -p no:cacheprovider > $out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -n 8 $out/${tag}_pytest.log
if [ $rc -gt 1 ]; then echo "pytest was killed or errored out: no further GPU step"; exit $rc; fi
unset CDV_TEST_LOG
for cfg in default stress; do
  timeout -k 10 400 python bench.py --steps 200 --warmup 20 --config $cfg > $out/${tag}_bench_$cfg.json 2> $out/${tag}_bench_$cfg.err
  brc=$?
  echo "bench $cfg rc=$brc"; tail -c 1800 $out/${tag}_bench_$cfg.json; tail -n 3 $out/${tag}_bench_$cfg.err
  if [ $brc -ne 0 ]; then exit $brc; fi
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_prof_$cfg -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --config $cfg --no-cpu-baseline --no-dropin > $GRAFT_REPO_ROOT/$out/${tag}_prof_$cfg.log 2>&1)
  prc=$?
  echo "rocprof $cfg rc=$prc"
  if [ $prc -ne 0 ]; then exit $prc; fi
  python scripts/kstats.py $out/${tag}_prof_$cfg > $out/${tag}_kernel_stats_$cfg.txt 2>&1; tail -n 22 $out/${tag}_kernel_stats_$cfg.txt
done
