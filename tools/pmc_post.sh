R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/post_sq -o p -- python $R/tools/post_times.py > $O/post_sq.log 2>&1
cd $R; python tools/rocpd_summary.py $(find $O/post_sq -name "*.db" | head -1) | grep "k_post_mc\|k_post_lnp1\|k_post_draw" | cut -c1-130
