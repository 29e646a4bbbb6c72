#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/gpu_ops_report.txt
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r3_call29.log
cp gpurun_out/gpu_ops_report.txt gpurun_out/r03_gpu_parity_report.txt
