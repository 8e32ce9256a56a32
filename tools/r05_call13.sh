#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python tools/two_stream_ab.py 2>&1 | grep -v amdgpu.ids
