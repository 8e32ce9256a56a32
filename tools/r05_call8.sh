#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
run() { timeout 300 python bench.py --steps 18 --warmup 1 --task-form $1 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('task form $1: value', d['value'], 'ms', d['ms_per_step'], 'serial', d['one_pass_at_a_time']['value'], 'timeouts', d['handoff_timeouts'], 'fallbacks', d['handoff_fallbacks'])
"; }
for f in 0 1 2 3 0; do run $f; done
