#!/bin/bash
# SQ counters of k_cluster on the bench's cluster workload:  tools/dev/cluster_pmc.sh <tag>
tag=${1:-clsq}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
W="python $R/bench.py --config 5 --steps 20 --warmup 2 --cpu-seconds 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS -d $O/${tag}_a -o p -- $W > $O/${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/${tag}_b -o p -- $W > $O/${tag}_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS -d $O/${tag}_c -o p -- $W > $O/${tag}_c.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS -d $O/${tag}_d -o p -- $W > $O/${tag}_d.log 2>&1
cd $R
for x in a b c d; do python tools/rocpd_summary.py $(find $O/${tag}_$x -name "*.db" | head -1) 2>&1 | grep -i "k_cluster<\|kernel\|counter" | head -12; done
