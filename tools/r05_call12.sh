#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c12
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/c12/tests_all.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/c12/tests_all.txt
tail -8 gpurun_out/c12/tests_all.txt | grep -v Warning
