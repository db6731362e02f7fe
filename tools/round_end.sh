#!/bin/bash
# End-of-round evidence on the GPU box (run through gpurun from the repo root):
#   tools/round_end.sh <tag> [commit]      e.g. tools/round_end.sh r03_v8 $(git rev-parse --short HEAD)
# the whole -m gpu suite, the default bench line, kernel traces of both end-to-end modes, then
# tools/profile_round.sh (kernel traces + FETCH_SIZE / WRITE_SIZE passes of the fit kernels).
tag=${1:-rXX}; commit=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/${tag}_gputest.log
python bench.py > $O/${tag}_bench_default.json 2> $O/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${tag}_np -o p -- python $R/tools/e2e_np.py 2048 > $O/${tag}_np.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${tag}_ph -o p -- python $R/tools/e2e_seq.py > $O/${tag}_ph.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/${tag}_np -name "*.db" | head -1) > $O/${tag}_kernel_trace_fit_end_to_end_numpy_rng.txt
python tools/rocpd_summary.py $(find $O/${tag}_ph -name "*.db" | head -1) > $O/${tag}_kernel_trace_fit_end_to_end_philox.txt
rm -rf $O/${tag}_np $O/${tag}_ph
bash tools/profile_round.sh $tag 128 $commit > $O/${tag}_profile_round.log 2>&1
rm -rf $O/${tag}_trace_cfg2 $O/${tag}_trace_cfg3 $O/${tag}_pmc_FETCH_SIZE_cfg2 $O/${tag}_pmc_FETCH_SIZE_cfg3 $O/${tag}_pmc_WRITE_SIZE_cfg2 $O/${tag}_pmc_WRITE_SIZE_cfg3
cat $O/${tag}_gputest.log; grep -v rocprofv3 $O/${tag}_np.log | tail -1; grep -v rocprofv3 $O/${tag}_ph.log | tail -1; tail -3 $O/${tag}_profile_round.log
