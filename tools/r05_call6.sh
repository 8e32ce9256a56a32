#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c6
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/c6/tests_all.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/c6/tests_all.txt
tail -8 gpurun_out/c6/tests_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c6/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/c6/smoke.txt
