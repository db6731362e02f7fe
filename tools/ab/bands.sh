#!/bin/bash
# same-box A/B of two builds at other band counts:  tools/ab/bands.sh "24 32" "base new"
nbs=${1:-"24 32"}; vars=${2:-"base new"}
for nb in $nbs; do for cfg in 2 3; do for v in $vars; do
  BRUTUS_AMD_LIB=$PWD/tools/ab/$v.so python bench.py --full-line --nfilt $nb --config $cfg --single-config --steps 6 --warmup 2 --repeats 3 --cpu-seconds 0 --e2e-stars 0 --no-survey-grid --no-sharp --no-cluster 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('%-8s nb$nb cfg$cfg %6d' % ('$v', round(d['value'])), {n:round(k[n]['avg_launch_ms'],3) for n in ('k_fflux','k_derive','k_pre32','k_top','k_sel_band','k_nomB') if n in k}, d.get('parity',{}).get('scan'))"
done; done; done
