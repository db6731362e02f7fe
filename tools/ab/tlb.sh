cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in 1 2 3 4 5 6; do
  rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE -d $R/gpurun_out/tlb_$r -o p -- python $R/tools/pmc_workload.py 2 128 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $(find $R/gpurun_out/tlb_$r -name "*.db" | head -1) | grep -E "k_emit|k_fflux<12, true, true>" | grep -E "UTCL|GRBM" | cut -c1-125
  rm -rf $R/gpurun_out/tlb_$r
  echo --
done
