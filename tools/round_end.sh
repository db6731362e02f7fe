R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/r03_v7_gputest.log
python bench.py > $O/r03_v7_bench_default.json 2> $O/r03_v7_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r7_np -o p -- python $R/tools/e2e_np.py 2048 > $O/r7_np.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/r7_ph -o p -- python $R/tools/e2e_seq.py > $O/r7_ph.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/r7_np -name "*.db" | head -1) > $O/r03_v7_kernel_trace_fit_end_to_end_numpy_rng.txt
python tools/rocpd_summary.py $(find $O/r7_ph -name "*.db" | head -1) > $O/r03_v7_kernel_trace_fit_end_to_end_philox.txt
rm -rf $O/r7_np $O/r7_ph
cat $O/r03_v7_gputest.log; tail -2 $O/r7_np.log; tail -2 $O/r7_ph.log
