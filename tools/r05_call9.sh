#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c9
timeout 900 python -m pytest tests -q -m gpu -k "lanes or in_flight" 2>&1 | grep -v Warning | tail -4
python bench.py --steps 20 --warmup 5 > gpurun_out/c9/bench.out 2> gpurun_out/c9/bench.err; echo "bench rc=$?"
grep "bench +" gpurun_out/c9/bench.err | grep -v "kernel \|warmup\|condition:" | tail -32
