#!/bin/bash
# the one-launch factorisation of the global BA: its tests, timing, in-kernel stamps, kernel trace.
# usage: scripts/gpu_r4_factor.sh TAG
tag=${1:-r4fac}; out=gpurun_out; export TMPDIR=/tmp; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "global or handoff or loop_closure or reference_scale" > $out/${tag}_pytest.log 2>&1
rc=$?
tail -5 $out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/bench_global_ba.py 300 96 > $out/${tag}_global_time.log 2>&1 || { tail -20 $out/${tag}_global_time.log; exit 1; }
grep -v amdgpu $out/${tag}_global_time.log
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_baf.py 300 96 2>&1 | grep -v amdgpu.ids | tee $out/${tag}_stamps.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_global -- python3 $GRAFT_REPO_ROOT/scripts/bench_global_ba.py 300 96 > $GRAFT_REPO_ROOT/$out/${tag}_global.log 2>&1)
python scripts/kstats.py $out/${tag}_global 16 > $out/${tag}_kernel_stats_global.txt 2>&1; head -6 $out/${tag}_kernel_stats_global.txt
