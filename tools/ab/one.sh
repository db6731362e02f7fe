#!/bin/bash
# one bench run of one variant:  tools/ab/one.sh <variant> <config> -> stars/s and the heavy kernels' launch times
v=$1; cfg=$2
BRUTUS_AMD_LIB=$PWD/tools/ab/$v.so python bench.py --full-line --config $cfg --single-config --steps 10 --warmup 3 --cpu-seconds 0 --e2e-stars 0 --no-survey-grid --no-sharp --no-cluster 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('%-8s cfg$cfg %6d' % ('$v', round(d['value'])), {n: round(k[n]['avg_launch_ms'], 3) for n in ('k_fflux', 'k_derive', 'k_pre32') if n in k}, round(d['roofline']['sum_of_kernels_ms_per_sub_batch'], 3))"
