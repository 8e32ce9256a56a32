#!/bin/bash
# round 5, gpurun call 2: activation-read microbenchmark, the conditioned beam test, and a full bench.py line
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
timeout 120 ./tools/probe_actread > gpurun_out/c2/probe_actread.txt 2>&1; echo "rc=$?" >> gpurun_out/c2/probe_actread.txt
cat gpurun_out/c2/probe_actread.txt
timeout 900 python -m pytest tests/test_wide_gpu.py -q -m gpu -k "beam5_winners_exact" > gpurun_out/c2/tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/c2/tests.txt
tail -4 gpurun_out/c2/tests.txt
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/c2/bench.json 2> gpurun_out/c2/bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/c2/bench.err
