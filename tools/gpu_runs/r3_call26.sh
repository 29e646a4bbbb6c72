#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/gpu_ops_report.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -k "x2_midsize" 2>&1 | tail -12 | tee gpurun_out/r3_call26.log
grep "x2" gpurun_out/gpu_ops_report.txt | tee -a gpurun_out/r3_call26.log
