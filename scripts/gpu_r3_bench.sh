#!/bin/bash
# the driver's bench command + a kernel trace of it, for one or more configs: scripts/gpu_r3_bench.sh TAG [configs...]
tag=$1; shift
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
for cfg in "${@:-default}"; do
  timeout -k 10 500 python bench.py --steps 200 --warmup 20 --config $cfg > $out/${tag}_bench_$cfg.json 2> $out/${tag}_bench_$cfg.err
  echo "bench $cfg rc=$?"; python - <<PY
import json
d=json.load(open("$out/${tag}_bench_$cfg.json"))
print({k: (round(v,1) if isinstance(v,float) else v) for k,v in d.items() if k in ("value","ms_per_step")}, "corr", round(d["roofline"]["avg_launch_ms"]*1e3,2), "us frac", round(d["roofline"]["frac"],3))
for k in ("dropin_fps","stress","stream_fps"):
    print(" ", k, d.get(k) and {kk: (round(vv,1) if isinstance(vv,float) else vv) for kk,vv in d[k].items() if kk not in ("what",)})
print("  cpu", d.get("cpu_baseline") and round(d["cpu_baseline"]["value"],3), "ate", d.get("ate_vs_oracle") and d["ate_vs_oracle"]["value"])
PY
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${tag}_prof_$cfg -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --config $cfg --no-cpu-baseline --no-dropin --no-extra > $GRAFT_REPO_ROOT/$out/${tag}_prof_$cfg.log 2>&1)
  echo "rocprof $cfg rc=$?"
  python scripts/kstats.py $out/${tag}_prof_$cfg 12 > $out/${tag}_kernel_stats_$cfg.txt 2>&1; cat $out/${tag}_kernel_stats_$cfg.txt
done
