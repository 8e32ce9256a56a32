#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/c3
timeout 240 ./tools/probe_engine > gpurun_out/c3/probe_engine_v3.txt 2>&1; echo "probe_engine rc=$?" >> gpurun_out/c3/probe_engine_v3.txt
cat gpurun_out/c3/probe_engine_v3.txt | cut -c1-330
