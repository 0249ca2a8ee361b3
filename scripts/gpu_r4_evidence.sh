#!/bin/bash
# the round's evidence in one call: GPU test suite, bench line + kernel trace (default), stress bench + trace, stream trace,
# in-kernel stamps (diagnostic build), and copies into profiles/ are made by hand afterwards
tag=${1:-r4final}
out=gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_r4_tests_bench.sh $tag || exit $?
bash scripts/gpu_r4_stress.sh $tag
bash scripts/gpu_r4_stream_prof.sh $tag
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_bam.py stress 2 > $out/${tag}_stamps_stress.log 2>&1
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps.py default 2>&1 | head -12 > $out/${tag}_stamps_corr.log
echo "stamps done"
