# round-4 GPU call 4: the per-round profile refresh (tools/final_profile.sh) + the conditioned large-v3 test (contention rounds)
R=$GRAFT_REPO_ROOT
cd $R
bash tools/final_profile.sh 2>&1 | tail -n 40
mkdir -p gpurun_out/c4
timeout 900 python -m pytest tests/test_wide_gpu.py -q -m gpu -k "conditioned_checkpoint and large" -s --durations=5 > gpurun_out/c4/tests.log 2>&1; echo "cond-large rc=$?"
grep -E "contention:|passed|failed|^E |s call" gpurun_out/c4/tests.log | cut -c1-900
