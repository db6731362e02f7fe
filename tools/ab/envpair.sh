#!/bin/bash
# same-box A/B of (library, environment) pairs, both configurations, round-robin:
#   tools/ab/envpair.sh <rounds> "<name>:<lib>[:VAR=val[,VAR=val...]] ..."   [configs]
n=${1:-2}; specs=${2:?specs}; cfgs=${3:-"2 3"}
for r in $(seq $n); do for cfg in $cfgs; do for sp in $specs; do
  name=${sp%%:*}; rest=${sp#*:}; lib=${rest%%:*}; envs=""
  [ "$rest" != "$lib" ] && envs=$(echo ${rest#*:} | tr ',' ' ')
  env $envs BRUTUS_AMD_LIB=$PWD/tools/ab/$lib.so python bench.py --full-line --config $cfg --single-config --steps 20 --warmup 4 --repeats 3 --cpu-seconds 0 --e2e-stars 0 --no-survey-grid --no-sharp --no-cluster --no-parity 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('%-10s cfg$cfg %6d' % ('$name', round(d['value'])), {n:round(k[n]['avg_launch_ms'],3) for n in ('k_fflux','k_derive','k_pre32','k_pre32_rows','k_top','k_sel_classify','k_surv_compact','k_select','k_k1probe','k_fflux_cont') if n in k}, round(d['roofline']['sum_of_kernels_ms_per_sub_batch'],3))"
done; done; done
