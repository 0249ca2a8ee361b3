#!/bin/bash
# A/B on the GPU box: the bench line + rocprofv3 kernel stats for each "NAME:ENV=VAL,ENV=VAL" variant given
# usage: scripts/gpu_r3_ab.sh TAG [config] variant...      e.g.  scripts/gpu_r3_ab.sh ab1 default base:CDV_CORR_STREAM=0 stream:CDV_CORR_STREAM=1
tag=$1; cfg=$2; shift 2
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  (
    IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
    timeout -k 10 300 python bench.py --steps 400 --warmup 20 --config $cfg --no-cpu-baseline --no-dropin --no-extra > $out/${tag}_${name}_bench.json 2> $out/${tag}_${name}_bench.err
    echo "[$name] bench rc=$?"; python - <<PY
import json
d=json.load(open("$out/${tag}_${name}_bench.json"))
print("[$name] fps %.0f  ms/step %.4f  corr event ms %.4f frac %.3f  stages %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], {k: round(v,1) for k,v in d["stages_us"].items()}))
PY
    cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_${name}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --config $cfg --no-cpu-baseline --no-dropin --no-extra > $GRAFT_REPO_ROOT/$out/${tag}_${name}_prof.log 2>&1
    echo "[$name] rocprof rc=$?"
    cd $GRAFT_REPO_ROOT && python scripts/kstats.py $out/${tag}_${name}_prof 9 > $out/${tag}_${name}_kstats.txt 2>&1; cat $out/${tag}_${name}_kstats.txt
  ) || exit 1
done
