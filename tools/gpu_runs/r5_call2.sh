#!/bin/bash
# round 5, call 2: the engine's frame-group mode (multi-map launches of the backward branches): bit-identity + quick regression
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call2.log
: > $L
echo "== frame groups ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 300 --timeout-method=thread -x \
  -k "frame_groups or pipelined_mode or stream_against_reference or round4 or context_export" 2>&1 | tail -25 | tee -a $L
