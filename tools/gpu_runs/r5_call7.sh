#!/bin/bash
# round 5, call 7: is the P | F | M group mode robust to the stream -> hardware-queue mapping?  Default bench (three call modes
# interleaved in one process) x hardware-queue counts, plus the group mode alone
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call7.log
: > $L
fmt='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f  samples %s  percall %s dropin %s layout %s streams %s" % (d["value"], d["samples"], d.get("one_frame_per_call") and d["one_frame_per_call"]["value"], d.get("dropin_surface") and d["dropin_surface"]["value"], d["config"].get("pipe_layout"), d.get("streams_ms_per_frame")))'
B="python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --full-json gpurun_out/_b_full.json"
run() {
  local name=$1; shift
  echo "== $name ==" | tee -a $L
  env "$@" timeout 240 $B $EXTRA > gpurun_out/_b.out 2> gpurun_out/_b.err
  tail -1 gpurun_out/_b.out | python -c "$fmt" 2>/dev/null | cut -c1-400 | tee -a $L || true
  if ! tail -1 gpurun_out/_b.out | grep -q '"value"'; then tail -4 gpurun_out/_b.err | cut -c1-400 | tee -a $L; fi
}
EXTRA=""
run "three modes interleaved, default" X=1
run "three modes interleaved, GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8
run "three modes interleaved, GPU_MAX_HW_QUEUES=2" GPU_MAX_HW_QUEUES=2
run "three modes interleaved, default (again)" X=1
EXTRA="--no-dropin"
run "group mode alone, default" X=1
run "group mode alone, GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8
run "group mode alone, keep pf_m" REFVSR_GROUP_KEEP_LAYOUT=1
run "group mode alone, default (again)" X=1
