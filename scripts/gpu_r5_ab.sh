#!/bin/bash
# tests named by $2 (pytest -k), then A/B of library variants named by $3.. on the default + stress workloads
tag=${1:-r5ab}; kexpr=${2:-}; shift 2
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
if [ -n "$kexpr" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "$kexpr" > $out/${tag}_pytest.log 2>&1
  rc=$?; echo "pytest rc=$rc"; tail -n 6 $out/${tag}_pytest.log
  if [ $rc -gt 1 ]; then exit $rc; fi
fi
for v in product "$@"; do
  if [ "$v" = product ]; then unset CDV_LIB; else export CDV_LIB=$PWD/cdv_slam_amd/libcdvslam_hip_$v.so; fi
  timeout -k 10 200 python scripts/exp_variants.py default stress >> $out/${tag}_variants.log 2>&1 || exit 1
done
unset CDV_LIB
cat $out/${tag}_variants.log | grep -v amdgpu.ids
