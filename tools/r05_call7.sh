#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c7
python bench.py --steps 20 --warmup 5 > gpurun_out/c7/bench.json 2> gpurun_out/c7/bench.err; echo "bench rc=$?"
grep -v "kernel \|warmup\|condition:" gpurun_out/c7/bench.err | grep "bench +\|Elapsed\|Maximum resident" | tail -40
