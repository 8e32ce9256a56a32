# tools/quick_profile.sh [tag] [extra bench args] — in-chain per-kernel times of the decode step (rocprofv3 kernel trace of a
# short bench.py run); summary -> gpurun_out/<tag>_kernel_stats.csv.  Run through gpurun.
R=$GRAFT_REPO_ROOT; TAG=${1:-quick}; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q
rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o q -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/${TAG}_prof.log 2>&1
K=$(find /tmp/prof_q -name "*_results.db" | head -n 1)
python $R/tools/prof_summary.py $K --grid --csv $R/gpurun_out/${TAG}_kernel_stats.csv > /dev/null 2>&1
python $R/tools/prof_summary.py $K --gaps --csv $R/gpurun_out/${TAG}_gaps.csv > /dev/null 2>&1
grep -E "timed|value" $R/gpurun_out/${TAG}_prof.log | cut -c1-200
head -n 24 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-170
