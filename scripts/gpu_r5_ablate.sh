#!/bin/bash
# ablations of the correlation kernel (diagnostic build): which phase's removal shortens the launch
export CDV_LIB=$PWD/cdv_slam_amd/libcdvslam_hip_stamps.so
for x in 0 1 2 16 128 256 512 3 130 384 899; do
  CDV_CORR_EXP=$x python scripts/exp_variants.py default 2>&1 | grep -v amdgpu | sed "s/^/exp=$x /"
done
