#!/bin/bash
# same-box A/B of two builds of the library: tools/ab/run.sh [config] [rounds]
cfg=${1:-2}; n=${2:-3}
for r in $(seq $n); do
  for v in base new; do
    BRUTUS_AMD_LIB=$PWD/tools/ab/$v.so python bench.py --full-line --config $cfg --single-config --steps 20 --warmup 4 --cpu-seconds 0 --e2e-stars 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$v', $cfg, round(d['value']), {n:round(k[n]['avg_launch_ms'],3) for n in ('k_fflux','k_emit','k_pre32') if n in k})"
  done
done
