#!/bin/bash
# round 5, call 12: the ADVICE r4 case on the GPU (context exchange, frame_num = 5, block start one frame before a reset frame);
# what the a-priori flagging margin of the arg-max (2^-10 instead of the empirical 2.5e-4) costs in the group mode
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call12.log
: > $L
echo "== exchange test ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 400 --timeout-method=thread -x -k "block_start_before or two_process" 2>&1 | tail -8 | tee -a $L
fmt='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f  samples %s  percall %s dropin %s streams %s" % (d["value"], d["samples"], d.get("one_frame_per_call") and d["one_frame_per_call"]["value"], d.get("dropin_surface") and d["dropin_surface"]["value"], d.get("streams_ms_per_frame")))'
B="python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --full-json gpurun_out/_b_full.json"
for m in default 9.8e-4 default 9.8e-4; do
  echo "== match margin $m ==" | tee -a $L
  if [ $m = default ]; then timeout 240 $B | python -c "$fmt" | cut -c1-400 | tee -a $L; else timeout 240 $B --match-margin $m | python -c "$fmt" | cut -c1-400 | tee -a $L; fi
done
