#!/bin/bash
# Same-box A/B of two BUILDS of the library on the bench workloads (run through gpurun from the repo root):
#   tools/ab_lib.sh <tag> <variant .so beside libbrutus_amd.so> [configs: "2 3"]
# The variant is copied over brutus_amd/libbrutus_amd.so for its runs (on the GPU box's scratch copy only).
tag=$1; var=$2; cfgs=${3:-"2 3"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
cp brutus_amd/libbrutus_amd.so /tmp/lib_default.so
: > $O/${tag}_ab.txt
for rep in 1 2; do
for v in default variant; do
  if [ $v = default ]; then cp /tmp/lib_default.so brutus_amd/libbrutus_amd.so; else cp $var brutus_amd/libbrutus_amd.so; fi
  for cfg in $cfgs; do
    python bench.py --config $cfg --single-config --no-survey-grid --no-sharp --no-cluster \
        --e2e-stars 0 --cpu-seconds 0 --steps 10 --warmup 2 --repeats 3 > $O/${tag}_line.json 2> $O/${tag}_err.txt || tail -5 $O/${tag}_err.txt >> $O/${tag}_ab.txt
    python - "$v" $cfg $rep >> $O/${tag}_ab.txt <<'PY'
import json, sys
d = json.load(open("bench_detail.json"))
k = d["roofline"].get("kernels", {})
top = sorted(k.items(), key=lambda kv: -kv[1]["avg_launch_ms"])[:6]
print("%-10s cfg %s rep %s  %9.1f stars/s  parity %s | %s" % (
    sys.argv[1], sys.argv[2], sys.argv[3], d["value"], (d.get("parity") or {}).get("sel_equal"),
    "  ".join("%s %.3f" % (n, e["avg_launch_ms"]) for n, e in top)))
PY
  done
done
done
cp /tmp/lib_default.so brutus_amd/libbrutus_amd.so
cat $O/${tag}_ab.txt
