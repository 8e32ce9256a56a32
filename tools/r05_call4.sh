#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c4
for f in 1 2 3 4; do
  timeout 300 python bench.py --steps 12 --warmup 1 --in-flight $f --no-cpu-baseline --no-extras --no-roofline 2> gpurun_out/c4/inflight_$f.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('in flight', d['config']['passes_in_flight'], 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'lanes equal', d.get('lanes_tokens_equal'))
"
done | tee gpurun_out/c4/inflight.txt
tail -3 gpurun_out/c4/inflight_2.err
