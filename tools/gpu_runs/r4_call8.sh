#!/bin/bash
# round 4, call 8: (a) s_memtime probe of the fused 48-channel block ("build the PROBE variant and show cycles");
# (b) two M streams for mid_channels = 24 under the two-stream layout of round 4 (round 3 measured it under P | F | M only);
# (c) stream layouts for RefVSR_MFID; (d) the engine-level test of the opt-in fused tail.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call8.log
: > $L
echo "== tests ==" | tee -a $L
timeout 400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider --timeout 240 --timeout-method=thread -k "round4 or resblock48" 2>&1 | tail -6 | tee -a $L
echo "== probe of resblock48 ==" | tee -a $L
timeout 200 python tools/probe_resblock48.py > gpurun_out/r04_resblock48_probe.txt 2>&1
cat gpurun_out/r04_resblock48_probe.txt | cut -c1-200 | tee -a $L
fmt='import sys,json
d=json.loads(sys.stdin.read())
print("value %.1f  samples %s  dropin %s  M %.2f P %.2f F %.2f" % (d["value"], d["samples"], d["dropin_surface"] and round(d["dropin_surface"]["value"],1), d["streams"]["median_pass"]["M_ms_per_call"], d["streams"]["median_pass"]["P_ms_per_call"], d["streams"]["median_pass"]["F_ms_per_call"]))'
B="python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --no-dropin"
run() {
  local name=$1; shift
  echo "== $name ==" | tee -a $L
  env "$@" timeout 240 $B $EXTRA > gpurun_out/_b.out 2> gpurun_out/_b.err
  tail -1 gpurun_out/_b.out | python -c "$fmt" 2>/dev/null | cut -c1-300 | tee -a $L || true
  if ! tail -1 gpurun_out/_b.out | grep -q '"value"'; then tail -4 gpurun_out/_b.err | cut -c1-400 | tee -a $L; fi
}
EXTRA=""
for round in 1 2; do
  run "small: default (round $round)" X=1
  run "small: REFVSR_PIPE_TWO_M=1 (round $round)" REFVSR_PIPE_TWO_M=1
  run "small: REFVSR_PIPE_TWO_M=1 REFVSR_BW_HEAD_BLOCKS=-1 (round $round)" REFVSR_PIPE_TWO_M=1 REFVSR_BW_HEAD_BLOCKS=-1
  run "small: REFVSR_PIPE_TWO_M=1 REFVSR_BW_HEAD_BLOCKS=6 (round $round)" REFVSR_PIPE_TWO_M=1 REFVSR_BW_HEAD_BLOCKS=6
done
EXTRA="--config config_RefVSR_MFID --steps 12 --warmup 3 --repeats 3"
for round in 1 2; do
  run "MFID: default (pf_m, two M) (round $round)" X=1
  run "MFID: REFVSR_PIPE_LAYOUT=pfm (round $round)" REFVSR_PIPE_LAYOUT=pfm
  run "MFID: REFVSR_PIPE_TWO_M=0 (round $round)" REFVSR_PIPE_TWO_M=0
done
