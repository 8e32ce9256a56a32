#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1200 python -m pytest tests/test_wide_gpu.py -q -m gpu -k "prefill_and_steps" 2>&1 | grep -v Warning | tail -6
