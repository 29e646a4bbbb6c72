#!/bin/bash
# round 4, call 17 (last seconds of the budget): the engine paths that pass through the lines the two-message hand-off touched
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call17.log
: > $L
timeout 75 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 60 --timeout-method=thread -x -k "stream_against_reference or pipelined or refvsr_ir or batch_and_api" > gpurun_out/_t.out 2>&1
grep -i -A14 "Traceback\|^E " gpurun_out/_t.out | head -40 | cut -c1-300 | tee -a $L
tail -3 gpurun_out/_t.out | tee -a $L
