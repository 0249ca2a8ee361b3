#!/bin/bash
# one GPU-box call: the GPU test suite only.  usage: scripts/gpu_r3_tests.sh TAG [pytest -k expression]
tag=${1:-r3}
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
echo "pytest rc=$rc"; grep -E "FAILED|ERROR|passed|failed" $out/${tag}_pytest.log | tail -n 15
exit $rc
