R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
O=$R/gpurun_out/final
python $R/__graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
python $R/bench.py --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
# per-kernel durations of the headline's schedule (one 24-row chain at a time: what bench.py's roofline object measures, kernel by kernel) ...
rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o k -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_k.log 2>&1
K=$(find /tmp/prof_k -name "*_results.db" | head -n 1); python $R/tools/prof_summary.py $K --grid --csv $O/kernel_stats.csv > $O/kernel_stats.txt 2>&1
# the non-headline legs (beam 5, word timestamps, base x 1, turbo x 32) in one kernel trace: where their time goes
rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o x -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof_x.log 2>&1
X=$(find /tmp/prof_x -name "*_results.db" | head -n 1); python $R/tools/prof_summary.py $X --grid --csv $O/kernel_stats_extras.csv > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python $R/bench.py --steps 3 --warmup 0 --sample-len 24 --no-cpu-baseline --no-roofline --no-extras > $O/prof_f.log 2>&1
F=$(find /tmp/prof_f -name "*_results.db" | head -n 1); python $R/tools/prof_summary.py $F --pmc --csv $O/pmc_fetch_size.csv > $O/pmc_f.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python $R/bench.py --steps 3 --warmup 0 --sample-len 24 --no-cpu-baseline --no-roofline --no-extras > $O/prof_w.log 2>&1
W=$(find /tmp/prof_w -name "*_results.db" | head -n 1); python $R/tools/prof_summary.py $W --pmc --csv $O/pmc_write_size.csv > $O/pmc_w.txt 2>&1
python $R/tools/pmc_traffic.py $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/pmc_traffic.json 24
tail -n 2 $O/smoke.log; tail -n 1 $O/bench.out | cut -c1-400; ls -la $O
# encoder GEMM / flash-attention probe (same box, same run): general kernel, rows kernel, K-loop floor, phase stamps
if [ -x $R/tools/probe_gemm ]; then
  { echo "== general kernel only (WH_GEMM_DEV=1)"; WH_GEMM_DEV=1 timeout 90 $R/tools/probe_gemm;
    echo "== rows kernel, one tile per workgroup (WH_GEMM_DEV=4)"; WH_GEMM_DEV=4 timeout 90 $R/tools/probe_gemm | head -n 3;
    echo "== product dispatch"; timeout 90 $R/tools/probe_gemm;
    [ -x $R/tools/probe_gemm_ne ] && { echo "== rows kernel, epilogue compiled out (-DWH_GEMM_PROBE_NOEPI)"; timeout 90 $R/tools/probe_gemm_ne | head -n 3; }
    [ -x $R/tools/probe_gemm_p ] && { echo "== phase stamps (-DWH_PROBE): 1 first K tile landed, 2 K loop done, 3 ring free, 5 stores issued, 7 next tile's first K tile landed"; timeout 90 $R/tools/probe_gemm_p | head -n 6; }
  } > $O/probe_gemm.txt 2>&1
fi
# flash-attention tile layout check (product layout vs plain, bit for bit) and the LDS counters of the encoder kernels
[ -x $R/tools/probe_flash_layout ] && timeout 60 $R/tools/probe_flash_layout > $O/probe_flash_layout.txt 2>&1
[ -x $R/tools/probe_gemm ] && bash $R/tools/pmc_lds.sh final > /dev/null 2>&1 && cp $R/gpurun_out/final_pmc_lds.csv $O/pmc_lds.csv
# matrix-core utilisation of the encoder kernels (SQ_VALU_MFMA_BUSY_CYCLES) on a short pass of the same bench
bash $R/tools/pmc_mfma.sh final > /dev/null 2>&1 && cp $R/gpurun_out/final_pmc_mfma.csv $R/gpurun_out/final_pmc_mfma_raw.csv $O/ 2>/dev/null
ls $O
