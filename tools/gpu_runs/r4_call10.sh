#!/bin/bash
# round 4, call 10: the three context-exchange tests again, with the workers' output
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call10.log
: > $L
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 300 --timeout-method=thread -s -k "context_export or exchange" > gpurun_out/_t.out 2>&1
grep -v "^$" gpurun_out/_t.out | grep -i -B2 -A25 "Traceback\|Error\|assert" | head -150 | cut -c1-300 | tee -a $L
tail -5 gpurun_out/_t.out | tee -a $L
