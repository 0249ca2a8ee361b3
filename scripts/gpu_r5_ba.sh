#!/bin/bash
# BA-focused GPU call: BA parity tests, window-BA stamps (diagnostic build), then the variants timing of the product build
tag=${1:-r5ba}
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "ba or BA or update or dropin or captured" > $out/${tag}_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 8 $out/${tag}_pytest.log
if [ $rc -gt 1 ]; then exit $rc; fi
CDV_LIB=$PWD/cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 300 python scripts/stamps_baw.py default 2 > $out/${tag}_stamps_default.log 2>&1 || exit 1
cat $out/${tag}_stamps_default.log
rm -f $out/${tag}_variants.log
for i in 1 2; do timeout -k 10 200 python scripts/exp_variants.py default stress >> $out/${tag}_variants.log 2>&1 || exit 1; done
grep -v amdgpu.ids $out/${tag}_variants.log
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dropin --no-extra > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
python -c "
import json;r=json.loads(open('$out/${tag}_bench.json').read().strip().split('\n')[-1]);print('bench', r['value'], r['ms_per_step'], r['stages_us'])"
