#!/bin/bash
out=gpurun_out; tag=${1:-r2s}
export TMPDIR=/tmp
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_baw.py default 1 > $out/${tag}_stamps1.log 2>&1 && \
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_baw.py default 2 > $out/${tag}_stamps2.log 2>&1
grep -v amdgpu.ids $out/${tag}_stamps1.log | head -12; grep -v amdgpu.ids $out/${tag}_stamps2.log
