#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
run() { timeout 300 python bench.py --steps 18 --warmup 1 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1: value', d['value'], 'ms', d['ms_per_step'], 'serial', d['one_pass_at_a_time']['value'])
"; }
run product
export WHISPER_AMD_LIB=whisper_amd/libwhisper_hip_dev.so
run dev-build
WH_GEMM_DEV=1 WH_GEMM_TILE=128 run "dev, encoder GEMMs on 128x128 tiles (4 waves, 64 KB LDS)"
WH_GEMM_DEV=1 run "dev, general kernel only (256x256)"
