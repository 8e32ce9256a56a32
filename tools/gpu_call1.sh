# round-4 GPU call 1: new probes, the new / changed tests, the bench, then the whole GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c1; mkdir -p $O
cd $R
timeout 120 tools/probe_mlp > $O/probe_mlp.txt 2>&1; echo "probe_mlp rc=$?"
PROBE_NO_OUT=1 timeout 180 tools/probe_fused > $O/probe_fused.txt 2>&1; echo "probe_fused rc=$?"
timeout 900 python -m pytest tests -q -m gpu -k "fused_step or handoff or survives or contention or conditioned" -x > $O/tests_new.log 2>&1; echo "tests_new rc=$?"
tail -n 5 $O/tests_new.log
timeout 600 python bench.py > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
tail -c 1500 $O/bench.out
timeout 900 python -m pytest tests -q -m gpu > $O/tests_all.log 2>&1; echo "tests_all rc=$?"
tail -n 8 $O/tests_all.log
cat $O/probe_mlp.txt
grep -E "us per link|published|fetched|stored" $O/probe_fused.txt | head -60
