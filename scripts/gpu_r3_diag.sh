#!/bin/bash
# diagnostics call: BA stamps (stress + default) and the host cost of the drop-in sequence.  usage: scripts/gpu_r3_diag.sh TAG
tag=${1:-r3d}; out=gpurun_out; export TMPDIR=/tmp
bash scripts/gpu_r3_stamps.sh $tag > /dev/null
grep -v amdgpu $out/${tag}_stamps_stress.log | head -48
timeout -k 10 200 python scripts/profile_dropin_host.py default 300 > $out/${tag}_dropin_host.log 2>&1
grep "dropin default" $out/${tag}_dropin_host.log
