#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/gpu_ops_report.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -k "ir_" 2>&1 | tail -6 | tee gpurun_out/r3_call25.log
grep "IR" gpurun_out/gpu_ops_report.txt | tee -a gpurun_out/r3_call25.log
