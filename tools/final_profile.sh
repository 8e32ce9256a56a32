R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
O=$R/gpurun_out/final
python $R/__graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
python $R/bench.py --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o k -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof_k.log 2>&1
K=$(find /tmp/prof_k -name "*_results.db" | head -n 1); python $R/tools/prof_summary.py $K --grid --csv $O/kernel_stats.csv > $O/kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python $R/bench.py --steps 1 --warmup 0 --sample-len 24 --no-cpu-baseline --no-roofline --no-extras > $O/prof_f.log 2>&1
F=$(find /tmp/prof_f -name "*_results.db" | head -n 1); python $R/tools/prof_summary.py $F --pmc --csv $O/pmc_fetch_size.csv > $O/pmc_f.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python $R/bench.py --steps 1 --warmup 0 --sample-len 24 --no-cpu-baseline --no-roofline --no-extras > $O/prof_w.log 2>&1
W=$(find /tmp/prof_w -name "*_results.db" | head -n 1); python $R/tools/prof_summary.py $W --pmc --csv $O/pmc_write_size.csv > $O/pmc_w.txt 2>&1
python $R/tools/pmc_traffic.py $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/pmc_traffic.json
tail -n 2 $O/smoke.log; tail -n 1 $O/bench.out | cut -c1-400; ls -la $O
