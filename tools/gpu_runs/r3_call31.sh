#!/bin/bash
# refvsr_conv32 (AlignedConv2d's 32-channel convs on the specialised kernel): op tests, full suite, bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call31.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "conv32 or conv24 or conv48" 2>&1 | tail -4 | tee -a $L
grep "conv32" gpurun_out/gpu_ops_report.txt | tee -a $L
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r03_gpu_parity_report.txt
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
B="python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin"
for i in 1 2 3; do
echo "conv32 specialised" | tee -a $L; timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "conv32 generic" | tee -a $L; REFVSR_NO_CONV32=1 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
