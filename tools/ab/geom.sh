#!/bin/bash
# bench geometry sweep on the GPU box: step size / sub-batch / streams -> stars/s (configs[1] and [2])
F="--single-config --steps 20 --warmup 4 --repeats 3 --cpu-seconds 0 --e2e-stars 0 --no-survey-grid --no-sharp --no-cluster --no-parity --no-kernel-timing"
for cfg in 2 3; do
for g in "512 128 3" "768 128 3" "512 256 2" "768 256 3" "1024 256 4" "1024 128 4" "512 64 4" "768 192 4"; do
  set -- $g
  v=$(python bench.py --config $cfg --batch $1 --sub-batch $2 --streams $3 $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['value_min']), round(d['value_max']))")
  echo "cfg$cfg batch $1 sub $2 streams $3: $v"
done; done
