#!/bin/bash
# round 5, gpurun call 1: the engine probe + the new parity tests
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
timeout 180 ./tools/probe_engine > gpurun_out/c1/probe_engine.txt 2>&1; echo "probe_engine rc=$?" >> gpurun_out/c1/probe_engine.txt
PROBE_L=1 timeout 120 ./tools/probe_engine > gpurun_out/c1/probe_engine_L1.txt 2>&1
timeout 1500 python -m pytest tests/test_wide_gpu.py -q -m gpu -k "beam5_winners_exact or turbo_dims_vs_oracle or alignment_conditioned" > gpurun_out/c1/tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/c1/tests.txt
tail -5 gpurun_out/c1/tests.txt
cat gpurun_out/c1/probe_engine.txt | head -60
