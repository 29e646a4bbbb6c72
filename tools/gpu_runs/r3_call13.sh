#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python tools/copy_sources.py 2>&1 | grep -v amdgpu.ids | tail -90 > gpurun_out/r3_call13_copy_sources.txt
tail -95 gpurun_out/r3_call13_copy_sources.txt
