# tools/pmc_mfma.sh [tag] — matrix-core utilisation evidence for the encoder kernels (north_star: "MFMA-busy against MI355X
# peak"): rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES on a short
# bench.py pass (counters only, no trace domains beyond --kernel-trace), summarised per kernel into
# gpurun_out/<tag>_pmc_mfma.csv.  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs).
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_m
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/prof_m -o m -- \
  python $R/bench.py --steps 1 --warmup 1 --sample-len 8 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/${TAG}_pmc_mfma.log 2>&1
M=$(find /tmp/prof_m -name "*_results.db" | head -n 1)
python $R/tools/prof_summary.py $M --pmc --csv $R/gpurun_out/${TAG}_pmc_mfma_raw.csv > /dev/null 2>&1
python $R/tools/pmc_mfma.py $R/gpurun_out/${TAG}_pmc_mfma_raw.csv $R/gpurun_out/${TAG}_pmc_mfma.csv
