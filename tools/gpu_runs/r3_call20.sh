#!/bin/bash
# streamed convs: four fragment sets in flight on the thin K-steps (REFVSR_CONV_NO_DEEP_RING=1 = two sets)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call20.log
: > $L
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "conv or spynet or vgg or match" 2>&1 | tail -4 | tee -a $L
for i in 1 2; do
timeout 200 python tools/bench_spynet.py 2>&1 | grep "spynet flow" | tee -a $L
REFVSR_CONV_NO_DEEP_RING=1 timeout 200 python tools/bench_spynet.py 2>&1 | grep "spynet flow" | sed 's/\[hi only\]/[hi only, two sets]/' | tee -a $L
done
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
B="python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin"
for i in 1 2 3; do
echo "four sets" | tee -a $L; timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "two sets" | tee -a $L; REFVSR_CONV_NO_DEEP_RING=1 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x -k "stream_against or full_size_against" 2>&1 | tail -3 | tee -a $L
