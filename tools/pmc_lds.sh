# tools/pmc_lds.sh [tag] — LDS pressure of the encoder kernels: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
# SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES on tools/probe_gemm (counters only, no trace domain beyond --kernel-trace);
# per-kernel averages -> gpurun_out/<tag>_pmc_lds.csv.  conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_l
timeout 100 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d /tmp/prof_l -o l -- \
  $R/tools/probe_gemm > $R/gpurun_out/${TAG}_pmc_lds.log 2>&1
L=$(find /tmp/prof_l -name "*_results.db" | head -n 1)
python $R/tools/prof_summary.py $L --pmc --csv $R/gpurun_out/${TAG}_pmc_lds.csv > /dev/null 2>&1
grep -E "rows_kernel|gemm_nt|flash" $R/gpurun_out/${TAG}_pmc_lds.csv | cut -c1-200 | head -40
