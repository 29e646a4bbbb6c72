#!/bin/bash
# round 4, call 15: REFVSR_WAVE_PRIO=1 (the second-dispatched half of every workgroup's waves at s_setprio 1 in resblock24 / conv24 /
# resblock48) against the default: op tests with the knob on (results must not depend on it), stand-alone blocks, the frame, RefVSR_MFID
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call15.log
: > $L
echo "== op tests with REFVSR_WAVE_PRIO=1 ==" | tee -a $L
REFVSR_WAVE_PRIO=1 timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider --timeout 100 --timeout-method=thread -k "resblock24 or conv24 or resblock48 or conv48 or conv_last or conf_alpha or shuffle" 2>&1 | tail -3 | tee -a $L
for P in 0 1; do
  echo "== stand-alone blocks, REFVSR_WAVE_PRIO=$P ==" | tee -a $L
  REFVSR_WAVE_PRIO=$P RB_ITERS=6 timeout 120 python tools/bench_resblock.py 2>&1 | grep "rb24 default " | tee -a $L
  REFVSR_WAVE_PRIO=$P timeout 120 python tools/bench_resblock48.py 2>&1 | grep -i "270x480\|540x960" | head -4 | tee -a $L
done
fmt='import sys,json
d=json.loads(sys.stdin.read())
print("value %.1f  samples %s" % (d["value"], d["samples"]))'
B="python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --no-dropin"
for round in 1 2; do
  for P in 0 1; do
    echo "== frame, REFVSR_WAVE_PRIO=$P (round $round) ==" | tee -a $L
    REFVSR_WAVE_PRIO=$P timeout 200 $B 2>/dev/null | tail -1 | python -c "$fmt" | tee -a $L
  done
done
for P in 0 1; do
  echo "== RefVSR_MFID, REFVSR_WAVE_PRIO=$P ==" | tee -a $L
  REFVSR_WAVE_PRIO=$P timeout 200 $B --config config_RefVSR_MFID --steps 12 --warmup 3 --repeats 3 2>/dev/null | tail -1 | python -c "$fmt" | tee -a $L
done
