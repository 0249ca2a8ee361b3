#!/bin/bash
# round 5, first GPU call: the GPU test suite, the bench line, then the capture diagnostic (one fault at most, last)
tag=${1:-r5a}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
rm -f $out/${tag}_ba_errors.jsonl
export CDV_TEST_LOG=$PWD/$out/${tag}_ba_errors.jsonl
timeout -k 10 1000 python -m pytest tests -m gpu -v --timeout 600 -p no:cacheprovider > $out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "FAILED|ERROR|passed|failed" $out/${tag}_pytest.log | tail -n 25
if [ $rc -gt 1 ]; then echo "pytest was killed or errored out: no further GPU step"; exit $rc; fi
timeout -k 10 600 python bench.py --steps 200 --warmup 20 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
brc=$?
echo "bench rc=$brc"; tail -c 600 $out/${tag}_bench.json; echo; tail -n 8 $out/${tag}_bench.err
if [ $brc -ne 0 ]; then exit $brc; fi
rm -f $out/r5_diag.log
bash scripts/gpu_r5_diag_capture.sh
