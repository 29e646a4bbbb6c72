#!/bin/bash
# full suite on the current tree + probe with per-XCD skews
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call16.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -5 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r03_gpu_parity_report.txt
PROBE_WAVES=8 timeout 200 python tools/probe_resblock24.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_probe_resblock24.txt | grep "==" | tee -a $L
