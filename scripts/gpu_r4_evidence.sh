#!/bin/bash
# the round's evidence in one call: GPU test suite, bench line + kernel trace (default), stress bench + trace, stream trace,
# in-kernel stamps (diagnostic build), the global BA (timing, trace, stamps of the one-launch factorisation, larger sizes);
# copies into profiles/ are made by hand afterwards
tag=${1:-r4final}
out=gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_r4_tests_bench.sh $tag || exit $?
bash scripts/gpu_r4_stress.sh $tag
bash scripts/gpu_r4_stream_prof.sh $tag
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_bam.py stress 2 > $out/${tag}_stamps_stress.log 2>&1
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps.py default 2>&1 | head -12 > $out/${tag}_stamps_corr.log
echo "stamps done"
timeout -k 10 300 python scripts/bench_global_ba.py 300 96 2>&1 | grep -v amdgpu > $out/${tag}_global_time.log
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_baf.py 300 96 2>&1 | grep -v amdgpu.ids > $out/${tag}_stamps_factor.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_global -- python3 $GRAFT_REPO_ROOT/scripts/bench_global_ba.py 300 96 > $GRAFT_REPO_ROOT/$out/${tag}_global.log 2>&1)
python scripts/kstats.py $out/${tag}_global 16 > $out/${tag}_kernel_stats_global.txt 2>&1
timeout -k 10 400 python scripts/check_global_sizes.py 600 32 2>&1 | grep -v amdgpu.ids > $out/${tag}_global_sizes.log
timeout -k 10 400 python scripts/check_global_sizes.py 1024 12 2>&1 | grep -v amdgpu.ids >> $out/${tag}_global_sizes.log
cat $out/${tag}_global_time.log $out/${tag}_global_sizes.log
