#!/bin/bash
# one GPU-box call: the GPU test suite, then (unless a step was killed) the bench line and a kernel trace of it
# usage: scripts/gpu_r2_tests_bench.sh TAG [pytest -k expression]
tag=${1:-r4}
kexpr=${2:-}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
rm -f $out/${tag}_ba_errors.jsonl
export CDV_TEST_LOG=$PWD/$out/${tag}_ba_errors.jsonl
if [ -n "$kexpr" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -v --timeout 600 -p no:cacheprovider -k "$kexpr" > $out/${tag}_pytest.log 2>&1
else
  timeout -k 10 1000 python -m pytest tests -m gpu -v --timeout 600 -p no:cacheprovider > $out/${tag}_pytest.log 2>&1
fi
rc=$?
echo "pytest rc=$rc"; tail -n 15 $out/${tag}_pytest.log
if [ $rc -gt 1 ]; then echo "pytest was killed or errored out: no further GPU step"; exit $rc; fi
timeout -k 10 600 python bench.py --steps 200 --warmup 20 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
brc=$?
echo "bench rc=$brc"; tail -c 1500 $out/${tag}_bench.json; tail -n 5 $out/${tag}_bench.err
if [ $brc -ne 0 ]; then exit $brc; fi
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-dropin --no-extra > $GRAFT_REPO_ROOT/$out/${tag}_prof.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT && python scripts/kstats.py $out/${tag}_prof > $out/${tag}_kernel_stats.txt 2>&1; tail -n 25 $out/${tag}_kernel_stats.txt
