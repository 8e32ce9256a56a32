# tools/lanes_timeouts.sh [runs, default 25] — VERDICT round 5 item 7: the bounded hand-off spins of the fused cross-attention launch
# (xattn8_kernel, X_MAX_SPINS = 8192) with THREE 8-row decode chains in flight, counted over fresh processes on a fresh box: each run
# is `bench.py --chain-batches 1 --in-flight 3` (3 chains of 8 rows, one host thread), 12 passes; prints value, time-outs, fallbacks.
R=$GRAFT_REPO_ROOT; N=${1:-25}; LANES_FORM=${LANES_FORM:-}      # LANES_FORM="--task-form 4": keep the fused cross attention in the lanes
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/lanes_timeouts.txt
echo "bench.py --chain-batches 1 --in-flight 3 $LANES_FORM --steps 12 --warmup 1 --no-cpu-baseline --no-extras --no-roofline, $N fresh processes" > $OUT
for i in $(seq 1 $N); do
  python $R/bench.py --chain-batches 1 --in-flight 3 $LANES_FORM --steps 12 --warmup 1 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('run %2d  value %7.1f  ms_per_pass %6.1f  one_at_a_time %6.1f  handoff_timeouts %d  fallbacks %d  groups_equal %s' % ($i, d['value'], d['ms_per_step'], d['one_pass_at_a_time']['value'], d['handoff_timeouts'], d['handoff_fallbacks'], d['chain_groups_equal_one_pass_at_a_time']))" >> $OUT
done
cat $OUT
