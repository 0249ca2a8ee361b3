#!/bin/bash
# quick GPU-box call: selected tests, bench (short), the stamp diagnostics
tag=${1:-r2q}
kexpr=${2:-ba}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -v --timeout 600 -p no:cacheprovider -k "$kexpr" > $out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -n 6 $out/${tag}_pytest.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dropin > $out/${tag}_bench.json 2> $out/${tag}_bench.err
brc=$?
echo "bench rc=$brc"; python -c "
import json
d=json.loads(open('$out/${tag}_bench.json').read().strip().splitlines()[-1])
print('fps', d['value'], 'ms', d['ms_per_step'], 'corr ms', d['roofline']['avg_launch_ms']); print(d['stages_us'])"
if [ $brc -ne 0 ]; then exit $brc; fi
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_baw.py > $out/${tag}_stamps.log 2>&1
echo "stamps rc=$?"; cat $out/${tag}_stamps.log
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-dropin > $GRAFT_REPO_ROOT/$out/${tag}_prof.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT && python scripts/kstats.py $out/${tag}_prof 8
