#!/bin/bash
# resblock24 on 16 x 32 tiles / sixteen waves (one workgroup per CU): tests, micro-benchmark, probe, bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call15.log
: > $L
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "resblock" 2>&1 | tail -4 | tee -a $L
timeout 300 python tools/bench_resblock.py 2>&1 | grep resblock | grep -v "4 waves\|lrelu" | tee -a $L
PROBE_WAVES=8 timeout 200 python tools/probe_resblock24.py 2>&1 | grep -v amdgpu.ids | grep -A14 "== LR (" | tee -a $L
PROBE_WAVES=16 timeout 200 python tools/probe_resblock24.py 2>&1 | grep -v amdgpu.ids | grep -A14 "== LR (" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x -k "stream_against or pipelined or two_phase" 2>&1 | tail -3 | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
for i in 1 2; do
echo "default (by size)" | tee -a $L
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "8 waves everywhere" | tee -a $L
REFVSR_RESBLOCK24_WAVES=8 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "16 waves everywhere" | tee -a $L
REFVSR_RESBLOCK24_WAVES=16 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
