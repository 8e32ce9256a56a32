#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c5
run() { timeout 300 python bench.py --steps 24 --warmup 1 --in-flight $1 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  in flight', d['config']['passes_in_flight'], 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'serial', (d.get('one_pass_at_a_time') or {}).get('value'), 'lanes equal', d.get('lanes_tokens_equal'))
"; }
{
echo "one stream per lane (task + torch ops)"; for f in 2 3 4 5; do run $f; done
} | tee gpurun_out/c5/inflight_sweep2.txt
