#!/bin/bash
# Same-box A/B of a library switch on the bench workloads (run through gpurun from the repo root):
#   tools/ab_env.sh <tag> <ENVVAR> <value A> <value B> [configs: "2 3"] [extra bench args]
# For each value and configuration: stars/s of the scan and the per-kernel durations (HIP events,
# sequential) of one sub-batch call, into gpurun_out/<tag>_ab.txt.
tag=$1; var=$2; a=$3; b=$4; cfgs=${5:-"2 3"}; shift 5
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/${tag}_ab.txt
for rep in 1 2; do
for v in $a $b; do
  for cfg in $cfgs; do
    env $var=$v python bench.py --config $cfg --single-config --no-survey-grid --no-sharp --no-cluster \
        --e2e-stars 0 --cpu-seconds 0 --steps 10 --warmup 2 --repeats 3 "$@" > $O/${tag}_line.json 2> $O/${tag}_err.txt || tail -5 $O/${tag}_err.txt >> $O/${tag}_ab.txt
    python - "$var=$v" $cfg $rep >> $O/${tag}_ab.txt <<'PY'
import json, sys
d = json.load(open("bench_detail.json"))
k = d["roofline"].get("kernels", {})
top = sorted(k.items(), key=lambda kv: -kv[1]["avg_launch_ms"])[:6]
print("%-28s cfg %s rep %s  %9.1f stars/s  parity %s | %s" % (
    sys.argv[1], sys.argv[2], sys.argv[3], d["value"],
    (d.get("parity") or {}).get("sel_equal"),
    "  ".join("%s %.3f" % (n, e["avg_launch_ms"]) for n, e in top)))
PY
  done
done
done
cat $O/${tag}_ab.txt
