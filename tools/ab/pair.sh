#!/bin/bash
# same-box A/B of two builds, both configurations, round-robin:  tools/ab/pair.sh [rounds] [variants]
n=${1:-2}; vars=${2:-"base new"}
for r in $(seq $n); do for cfg in 2 3; do for v in $vars; do
  BRUTUS_AMD_LIB=$PWD/tools/ab/$v.so python bench.py --full-line --config $cfg --single-config --steps 20 --warmup 4 --repeats 3 --cpu-seconds 0 --e2e-stars 0 --no-survey-grid --no-sharp --no-cluster --no-parity 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('%-8s cfg$cfg %6d' % ('$v', round(d['value'])), {n:round(k[n]['avg_launch_ms'],3) for n in ('k_fflux','k_derive','k_pre32','k_top','k_sel_classify','k_surv_compact','k_select','k_k1probe','k_fflux_cont') if n in k}, round(d['roofline']['sum_of_kernels_ms_per_sub_batch'],3))"
done; done; done
