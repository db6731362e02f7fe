#!/bin/bash
# Effective shader clock per kernel: GRBM_GUI_ACTIVE (cycles the GPU was busy) / kernel duration.
#   tools/pmc_clock.sh <tag> [config] [batch]
tag=${1:-clk}; cfg=${2:-2}; B=${3:-128}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/${tag}_clk -o p -- python $R/tools/pmc_workload.py $cfg $B > $O/${tag}_clk.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/${tag}_clk -name "*.db" | head -1) > $O/${tag}_clk.txt
rm -rf $O/${tag}_clk
cat $O/${tag}_clk.txt
