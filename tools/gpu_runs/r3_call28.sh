#!/bin/bash
# conv24: x tiles fetched two tiles ahead (REFVSR_CONV24_PF1=1: one): op tests, micro-benchmark, bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call28.log
: > $L
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_torch_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "conv" 2>&1 | tail -3 | tee -a $L
timeout 200 python tools/bench_conv24.py 2>&1 | grep "^conv24" | tee -a $L
REFVSR_CONV24_PF1=1 timeout 200 python tools/bench_conv24.py 2>&1 | grep "^conv24" | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
B="python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin"
for i in 1 2 3; do
echo "two tiles ahead" | tee -a $L; timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "one tile ahead (24 -> 24 only)" | tee -a $L; REFVSR_CONV24_PF1=1 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x -k "stream_against or full_size_against or mfid or 8k" 2>&1 | tail -3 | tee -a $L
