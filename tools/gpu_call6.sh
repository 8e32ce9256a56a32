# round-4 GPU call 6: the final bench log + the GPU suite (alignment tests validated in call 3, unchanged since)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c6; mkdir -p $O
cd $R
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
tail -c 600 $O/bench.out; echo
timeout 900 python -m pytest tests -q -m gpu --durations=8 --deselect tests/test_wide_gpu.py::test_alignment_conditioned_fp16_equals_fp32 > $O/tests_all.log 2>&1; echo "tests rc=$?"
tail -n 16 $O/tests_all.log | cut -c1-200
