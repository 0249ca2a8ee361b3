#!/bin/bash
# round 5 evidence in one call: the GPU test suite, the bench line, kernel traces (default / stress / stream), in-kernel
# stamps of the window BA (diagnostic build), the captured-memset micro check.  Copies into profiles/ are made afterwards.
tag=${1:-r5final}
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $out/${tag}_ba_errors.jsonl
export CDV_TEST_LOG=$PWD/$out/${tag}_ba_errors.jsonl
timeout -k 10 1100 python -m pytest tests -m gpu -v --timeout 600 -p no:cacheprovider > $out/${tag}_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "FAILED|ERROR|passed|failed" $out/${tag}_pytest.log | tail -n 12
if [ $rc -gt 1 ]; then echo "pytest was killed or errored out: no further GPU step"; exit $rc; fi
unset CDV_TEST_LOG
timeout -k 10 600 python bench.py --steps 200 --warmup 20 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
brc=$?; echo "bench rc=$brc"; tail -n 4 $out/${tag}_bench.err
if [ $brc -ne 0 ]; then exit $brc; fi
python scripts/micro/memset_node.py > $out/${tag}_memset_node.log 2>&1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/${tag}_prof -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-dropin --no-extra > $R/$out/${tag}_prof.log 2>&1); echo "rocprof default rc=$?"
python scripts/kstats.py $out/${tag}_prof > $out/${tag}_kernel_stats_default.txt 2>&1; head -8 $out/${tag}_kernel_stats_default.txt
timeout -k 10 300 python bench.py --config stress --no-cpu-baseline --no-dropin --no-extra --windows 3 > $out/${tag}_bench_stress.json 2> $out/${tag}_bench_stress.err || exit 1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/${tag}_prof_stress -- python3 $R/bench.py --config stress --steps 60 --warmup 10 --no-cpu-baseline --no-dropin --no-extra --windows 1 > $R/$out/${tag}_prof_stress.log 2>&1); echo "rocprof stress rc=$?"
python scripts/kstats.py $out/${tag}_prof_stress 10 > $out/${tag}_kernel_stats_stress.txt 2>&1; head -8 $out/${tag}_kernel_stats_stress.txt
python scripts/bench_stream.py 600 > $out/${tag}_stream.log 2>&1; tail -2 $out/${tag}_stream.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/${tag}_stream_prof -- python3 $R/scripts/bench_stream.py 300 > $R/$out/${tag}_stream_prof.log 2>&1); echo "rocprof stream rc=$?"
python scripts/kstats.py $out/${tag}_stream_prof 30 > $out/${tag}_kernel_stats_stream.txt 2>&1; head -12 $out/${tag}_kernel_stats_stream.txt
[ cdv_slam_amd/libcdvslam_hip_stamps.so -nt cdv_slam_amd/csrc/ba.hip ] || echo "WARNING: libcdvslam_hip_stamps.so is older than the sources (make -C cdv_slam_amd/csrc STAMPS=1)"
CDV_LIB=$PWD/cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_baw.py default 2 > $out/${tag}_stamps_default.log 2>&1
CDV_LIB=$PWD/cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_bam.py stress 2 > $out/${tag}_stamps_stress.log 2>&1
CDV_LIB=$PWD/cdv_slam_amd/libcdvslam_hip_stamps.so timeout -k 10 200 python scripts/stamps_backsolve.py > $out/${tag}_stamps_backsolve.log 2>&1
python scripts/bench_global_ba.py > $out/${tag}_global_time.log 2>&1
echo "stamps done"; grep -A6 "solver wave" $out/${tag}_stamps_default.log
# the reference's call sequence through the drop-in names: host time per call (compiled bookkeeping on / off) and its kernels
python scripts/profile_dropin_host.py default 400 2>&1 | grep -v amdgpu.ids > $out/${tag}_dropin_host.log
CDV_DROPIN_FAST=0 python scripts/profile_dropin_host.py default 400 2>&1 | grep -v amdgpu.ids > $out/${tag}_dropin_host_python.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/${tag}_prof_dropin -- python3 $R/scripts/run_dropin.py default 300 > $R/$out/${tag}_prof_dropin.log 2>&1); echo "rocprof dropin rc=$?"
python scripts/kstats.py $out/${tag}_prof_dropin 24 > $out/${tag}_kernel_stats_dropin.txt 2>&1
head -1 $out/${tag}_dropin_host.log; head -1 $out/${tag}_dropin_host_python.log
