#!/bin/bash
# refvsr_conv_shuffle2 (specialised C -> 4 C conv + pixel shuffle): op tests, stream tests, bench A/B against the generic kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call27.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "shuffle" 2>&1 | tail -6 | tee -a $L
grep "conv_shuffle2" gpurun_out/gpu_ops_report.txt | tee -a $L
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x -k "stream_against or full_size_against or mfid" 2>&1 | tail -3 | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
B="python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin"
for i in 1 2 3; do
echo "specialised shuffle conv" | tee -a $L; timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "generic shuffle conv" | tee -a $L; REFVSR_NO_CONV_SHUFFLE2=1 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
echo "MFID specialised" | tee -a $L; timeout 400 python bench.py --config config_RefVSR_MFID --steps 20 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "MFID generic" | tee -a $L; REFVSR_NO_CONV_SHUFFLE2=1 timeout 400 python bench.py --config config_RefVSR_MFID --steps 20 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
