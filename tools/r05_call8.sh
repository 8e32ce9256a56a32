#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c8
timeout 900 python -m pytest tests/test_wide_gpu.py -q -m gpu -k "three_lanes" -s 2>&1 | grep -v Warning | tail -25
timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('in flight', d['config']['passes_in_flight'], 'value', d['value'], 'timeouts', d['handoff_timeouts'], 'fallbacks', d['handoff_fallbacks'])
"
