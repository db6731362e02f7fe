#!/bin/bash
# A/B the hot path on one GPU box: tools/ab_bench.sh <base.so> [bench args...]
# (alternates the baseline library and the in-tree build, two rounds each)
base=$1; shift
show='import json,sys; d=json.loads(sys.stdin.read()); print("%8.0f stars/s" % d["value"], {k: round(v, 3) for k, v in d["roofline"]["all_kernels_ms"].items()})'
for r in 1 2; do
  echo -n "base: "; BRUTUS_AMD_LIB=$base python bench.py --e2e-stars 0 --cpu-seconds 0 "$@" 2>&1 | tail -1 | python -c "$show"
  echo -n "new : "; python bench.py --e2e-stars 0 --cpu-seconds 0 "$@" 2>&1 | tail -1 | python -c "$show"
done
