#!/bin/bash
# parity block of the bench (4 stars against the C oracle) for A/B builds:  tools/ab/parity.sh "<lib> ..." [config]
for v in $1; do
  BRUTUS_AMD_LIB=$PWD/tools/ab/$v.so python bench.py --config ${2:-2} --single-config --steps 6 --warmup 2 --repeats 1 --cpu-seconds 0 --e2e-stars 0 --no-survey-grid --no-sharp --no-cluster 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), d.get('parity'))"
done
