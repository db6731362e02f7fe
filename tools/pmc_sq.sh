#!/bin/bash
# SQ counters of the hot-path kernels (issue efficiency, wait cycles):
#   tools/pmc_sq.sh <tag> [config] [batch]
tag=${1:-sq}; cfg=${2:-2}; B=${3:-64}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS -d $O/${tag}_a -o p -- python $R/tools/pmc_workload.py $cfg $B > $O/${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/${tag}_b -o p -- python $R/tools/pmc_workload.py $cfg $B > $O/${tag}_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS -d $O/${tag}_c -o p -- python $R/tools/pmc_workload.py $cfg $B > $O/${tag}_c.log 2>&1
cd $R
for x in a b c; do python tools/rocpd_summary.py $(find $O/${tag}_$x -name "*.db" | head -1) > $O/${tag}_$x.txt; rm -rf $O/${tag}_$x; done   # (raw databases: gpurun_out/ returns at most 64 MiB)
cat $O/${tag}_a.txt $O/${tag}_b.txt $O/${tag}_c.txt
