#!/bin/bash
# full GPU suite with plain fp16 weights in SPyNet's streamed convs: which bars move
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call12.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -25 | tee -a $L
grep -i "psnr\|spynet\|flow=" gpurun_out/gpu_ops_report.txt | awk '{print}' | sort -t= -k6 | tail -60 > gpurun_out/r3_call12_report.txt
cp gpurun_out/gpu_ops_report.txt gpurun_out/r3_call12_full_report.txt
