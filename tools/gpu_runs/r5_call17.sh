#!/bin/bash
# round 5, call 17: eight flows per SPyNet pass in group mode (was two passes of four): tests + same-box bench A/B against the chunk of 4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call17.log
: > $L
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 300 --timeout-method=thread -x -k "batched_conv_and_spynet or spynet or frame_groups_are or stream_against_reference" 2>&1 | tail -4 | tee -a $L
fmt='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f  samples %s  streams %s" % (d["value"], d["samples"], d.get("streams_ms_per_frame")))'
B="python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs --full-json gpurun_out/_b.json"
for i in 1 2; do
  echo "== 8 flows per pass ==" | tee -a $L
  timeout 240 $B | python -c "$fmt" | tee -a $L
  echo "== 4 flows per pass (REFVSR_SPYNET_CHUNK=4) ==" | tee -a $L
  REFVSR_SPYNET_CHUNK=4 timeout 240 $B | python -c "$fmt" | tee -a $L
done
