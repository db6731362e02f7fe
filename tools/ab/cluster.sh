#!/bin/bash
# same-box A/B of builds on the cluster-mode line:  tools/ab/cluster.sh "u1 u2 u4"
for r in 1 2; do for v in $1; do
  BRUTUS_AMD_LIB=$PWD/tools/ab/$v.so python bench.py --config 5 --steps 300 --warmup 10 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-6s' % '$v', round(d['value']), round(d['ms_per_step'],3), 'plug-in', round(d['plugin_ms_per_step'],3), 'revisit', round(d['revisited_table_evaluations_per_s']), 'k_cluster ms', round(d['roofline']['avg_launch_ms'],4))"
done; done
