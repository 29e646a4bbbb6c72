#!/bin/bash
# SPyNet: plain fp16 weights in the streamed 7x7 convs vs hi + lo, 8 waves x 2 pixel groups vs 4 x 4; bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call11.log
: > $L
REFVSR_SPYNET_HILO=1 SPYNET_DUMP=/tmp/flow_hilo.pt timeout 200 python tools/bench_spynet.py 2>&1 | grep "spynet flow" | tee -a $L
SPYNET_CMP=/tmp/flow_hilo.pt timeout 200 python tools/bench_spynet.py 2>&1 | grep "spynet flow" | tee -a $L
REFVSR_SPYNET_HILO=1 REFVSR_CONV_NO_NW8=1 SPYNET_CMP=/tmp/flow_hilo.pt timeout 200 python tools/bench_spynet.py 2>&1 | grep "spynet flow" | tee -a $L
REFVSR_CONV_NO_NW8=1 SPYNET_CMP=/tmp/flow_hilo.pt timeout 200 python tools/bench_spynet.py 2>&1 | grep "spynet flow" | tee -a $L
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -8 | tee -a $L
grep -i "psnr\|spynet" gpurun_out/gpu_ops_report.txt | head -20 | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
for i in 1 2; do
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
REFVSR_SPYNET_HILO=1 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
