#!/bin/bash
# in-kernel cycle stamps of both BA paths (diagnostic build: make -C cdv_slam_amd/csrc STAMPS=1).  usage: scripts/gpu_r3_stamps.sh TAG
out=gpurun_out; tag=${1:-r3s}
export TMPDIR=/tmp
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_bam.py stress 2 > $out/${tag}_stamps_stress.log 2>&1 && \
CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_baw.py default 2 > $out/${tag}_stamps_default.log 2>&1
grep -v amdgpu.ids $out/${tag}_stamps_stress.log; grep -v amdgpu.ids $out/${tag}_stamps_default.log
