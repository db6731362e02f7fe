#!/bin/bash
# End-of-round evidence on the GPU box, one gpurun call from the repo root:
#   tools/round_end.sh <tag> [commit]      e.g. tools/round_end.sh r05_v5 $(git rev-parse --short HEAD)
# 1. the whole -m gpu suite;  2. tools/profile_round.sh (kernel traces of configs[1] / [2] + FETCH_SIZE /
# WRITE_SIZE passes) and the SQ / clock passes;  3. the two tables bench.py reads (profiles/pmc_traffic.json,
# profiles/sq_valu.json) regenerated from those passes and the ISA of the current sources -- written to
# profiles/ on the box AND to gpurun_out/ (only that directory travels back: copy them over afterwards);
# 4. the default bench line;  5. kernel traces of both end-to-end modes;  6. smoke().
tag=${1:-rXX}; commit=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/${tag}_gputest.log
bash tools/profile_round.sh $tag 128 $commit > $O/${tag}_profile_round.log 2>&1
rm -rf $O/${tag}_trace_cfg2 $O/${tag}_trace_cfg3 $O/${tag}_pmc_FETCH_SIZE_cfg2 $O/${tag}_pmc_FETCH_SIZE_cfg3 $O/${tag}_pmc_WRITE_SIZE_cfg2 $O/${tag}_pmc_WRITE_SIZE_cfg3
bash tools/pmc_sq.sh ${tag}_sq_cfg2 2 128 > /dev/null 2>&1
bash tools/pmc_sq.sh ${tag}_sq_cfg3 3 128 > /dev/null 2>&1
bash tools/pmc_clock.sh ${tag}_cfg2 2 128 > /dev/null 2>&1
# the static ISA of both translation units (12-band instantiations: what the workloads run)
I=/tmp/isa_round; mkdir -p $I; ( cd $I
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -Wno-unused-value -save-temps -DBRUTUS_DEV_NB12_ONLY $R/brutus_amd/csrc/brutus_kernels.hip -o x.o > /dev/null 2>&1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -Wno-unused-value -save-temps -fno-slp-vectorize -DBRUTUS_DEV_NB12_ONLY $R/brutus_amd/csrc/pre32s_unit.hip -o y.o > /dev/null 2>&1 )
S="$I/brutus_kernels-hip-amdgcn-amd-amdhsa-gfx950.s $I/pre32s_unit-hip-amdgcn-amd-amdhsa-gfx950.s"
export PMC_COMMIT=$commit
rm -f $O/${tag}_sq_valu.json
python tools/sq_to_json.py $O/${tag}_sq_cfg2_a.txt 2 128 3 $O/${tag}_sq_valu.json $S > $O/${tag}_tables.log 2>&1
python tools/sq_to_json.py $O/${tag}_sq_cfg3_a.txt 3 128 3 $O/${tag}_sq_valu.json $S >> $O/${tag}_tables.log 2>&1
cp $O/${tag}_sq_valu.json profiles/sq_valu.json
cp $O/${tag}_pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $O/${tag}_bench_default.json 2> $O/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${tag}_np -o p -- python $R/tools/e2e_np.py 2048 > $O/${tag}_np.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${tag}_ph -o p -- python $R/tools/e2e_seq.py > $O/${tag}_ph.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/${tag}_np -name "*.db" | head -1) > $O/${tag}_kernel_trace_fit_end_to_end_numpy_rng.txt
python tools/rocpd_summary.py $(find $O/${tag}_ph -name "*.db" | head -1) > $O/${tag}_kernel_trace_fit_end_to_end_philox.txt
rm -rf $O/${tag}_np $O/${tag}_ph
python -c "import __graft_entry__ as g; g.smoke()" > $O/${tag}_smoke.log 2>&1
cat $O/${tag}_gputest.log $O/${tag}_tables.log; tail -1 $O/${tag}_smoke.log; tail -c 200 $O/${tag}_bench.err
