#!/bin/bash
# GPU-box call for the 10 < N <= 32 BA path: its tests, the stress bench line, stamps, kernel trace
tag=${1:-r2m}
kexpr=${2:-"ba or ate or status"}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
export CDV_TEST_LOG=$PWD/$out/${tag}_ba_errors.jsonl; rm -f $CDV_TEST_LOG
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "$kexpr" > $out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -n 8 $out/${tag}_pytest.log
if [ $rc -gt 1 ]; then exit $rc; fi
unset CDV_TEST_LOG
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --config stress --no-cpu-baseline --no-dropin > $out/${tag}_bench.json 2> $out/${tag}_bench.err
brc=$?
echo "bench rc=$brc"; python -c "
import json
d=json.loads(open('$out/${tag}_bench.json').read().strip().splitlines()[-1])
print('fps', d['value'], 'ms', d['ms_per_step'], 'corr ms', d['roofline']['avg_launch_ms']); print(d['stages_us'])"
if [ $brc -ne 0 ]; then exit $brc; fi
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_bam.py stress 2 > $out/${tag}_stamps.log 2>&1
echo "stamps rc=$?"; cat $out/${tag}_stamps.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --config stress --no-cpu-baseline --no-dropin > $GRAFT_REPO_ROOT/$out/${tag}_prof.log 2>&1)
echo "rocprof rc=$?"
python scripts/kstats.py $out/${tag}_prof 8
