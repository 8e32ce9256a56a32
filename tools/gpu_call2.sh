# round-4 GPU call 2: the new / changed tests, the bench with its new parity legs, then the whole GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c2; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu -k "fused_step or handoff or survives or contention or conditioned or encoder_mid or mid_width or wide_encoder" > $O/tests_new.log 2>&1; echo "tests_new rc=$?"
tail -n 12 $O/tests_new.log
timeout 900 python bench.py > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
tail -n 25 $O/bench.err | cut -c1-250
tail -c 3000 $O/bench.out
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_wide_gpu.py::test_conditioned_checkpoint_token_exact_224_steps --deselect tests/test_wide_gpu.py::test_alignment_conditioned_fp16_equals_fp32 --deselect tests/test_wide_gpu.py::test_fused_step_kernels_under_contention > $O/tests_all.log 2>&1; echo "tests_all rc=$?"
tail -n 15 $O/tests_all.log
