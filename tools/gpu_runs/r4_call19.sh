#!/bin/bash
# round 4, call 19: the op test of refvsr_conv1x1_f32 (call 18 tripped over a missing import in the test) + the matching / fixture /
# full-size parity tests with the kernel on
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call19.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 48 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 40 --timeout-method=thread -k "conv1x1 or feature_match or match_fused or stream_against_reference_fixture or full_size" > gpurun_out/_t.out 2>&1
grep -i -A10 "Traceback\|^E " gpurun_out/_t.out | head -30 | cut -c1-300 | tee -a $L
tail -2 gpurun_out/_t.out | tee -a $L
grep "conv1x1\|full-size\|idx" gpurun_out/gpu_ops_report.txt | cut -c1-200 | head -14 | tee -a $L
