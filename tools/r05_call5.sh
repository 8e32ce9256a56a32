#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c5
run() { timeout 300 python bench.py --steps 24 --warmup 1 --in-flight $1 --lane-priority $2 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  in flight', d['config']['passes_in_flight'], 'priority $2', 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'serial', (d.get('one_pass_at_a_time') or {}).get('value'))
"; }
{ run 3 0; run 3 -1; run 4 -1; run 3 0; } | tee gpurun_out/c5/inflight_priority.txt
