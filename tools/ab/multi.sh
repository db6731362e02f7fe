#!/bin/bash
# same-box A/B of several builds:  tools/ab/multi.sh "<variants>" [rounds] [configs]
# prints stars/s and the heavy kernels' mean launch times per variant, round-robin.
vars=${1:?variants}; n=${2:-2}; cfgs=${3:-"2 3"}
for r in $(seq $n); do
  for cfg in $cfgs; do
    for v in $vars; do
      BRUTUS_AMD_LIB=$PWD/tools/ab/$v.so python bench.py --full-line --config $cfg --single-config --steps 12 --warmup 3 --cpu-seconds 0 --e2e-stars 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('%-12s cfg$cfg %6d' % ('$v', round(d['value'])), {n:round(k[n]['avg_launch_ms'],3) for n in ('k_fflux','k_fflux_cont','k_derive','k_pre32','k_select','k_sel_classify','k_surv_compact','k_top') if n in k}, round(d['roofline']['sum_of_kernels_ms_per_sub_batch'],3))"
    done
  done
done
