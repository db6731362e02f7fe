#!/bin/bash
# One profiling pass on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [batch] [commit]     e.g. tools/profile_round.sh r01_v4 128 $(git rev-parse --short HEAD)
# kernel trace of the bench (configs[1] and [2], one stream), PMC FETCH_SIZE / WRITE_SIZE passes for
# configs[1] and configs[2] (separate runs, kernel-trace only), summaries into
# gpurun_out/<tag>_*.txt; raw databases stay in gpurun_out/.
tag=${1:-prof}
B=${2:-128}
export PMC_COMMIT=${3:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# --streams 1: with two streams the kernels of two batches overlap and every
# duration in the trace is stretched; one stream gives the per-launch durations
# that bench.py's own HIP-event pass (also sequential) must agree with
for cfg in 2 3; do
  rocprofv3 --kernel-trace --stats -d $O/${tag}_trace_cfg${cfg} -o t -- python $R/bench.py --config $cfg \
      --steps 4 --warmup 1 --batch 512 --sub-batch $B --streams 1 --cpu-seconds 0 --e2e-stars 0 --single-config --no-survey-grid --no-sharp --no-cluster \
      > $O/${tag}_bench_under_rocprof_cfg${cfg}_b${B}.json 2> $O/${tag}_trace_cfg${cfg}.log
done
for cfg in 2 3; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d $O/${tag}_pmc_${ctr}_cfg${cfg} -o p -- \
        python $R/tools/pmc_workload.py $cfg $B > $O/${tag}_pmc_${ctr}_cfg${cfg}.log 2>&1
  done
done
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
for cfg in 2 3; do
  python tools/rocpd_summary.py $(db ${tag}_trace_cfg${cfg}) > $O/${tag}_kernel_trace_cfg${cfg}_b${B}.txt
  python tools/rocpd_summary.py $(db ${tag}_pmc_FETCH_SIZE_cfg${cfg}) > $O/${tag}_pmc_fetch_cfg${cfg}_b${B}.txt
  python tools/rocpd_summary.py $(db ${tag}_pmc_WRITE_SIZE_cfg${cfg}) > $O/${tag}_pmc_write_cfg${cfg}_b${B}.txt
  python tools/pmc_to_json.py $(db ${tag}_pmc_FETCH_SIZE_cfg${cfg}) $(db ${tag}_pmc_WRITE_SIZE_cfg${cfg}) $cfg $B 268435456 $O/${tag}_pmc_traffic.json
done
tail -1 $O/${tag}_bench_under_rocprof_cfg2_b${B}.json | cut -c1-400
head -12 $O/${tag}_kernel_trace_cfg2_b${B}.txt
