#!/bin/bash
# refvsr_conv_shuffle2 with the activation epilogue (upsample2): full suite, bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call30.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r03_gpu_parity_report.txt
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms", [ (k["kernel"][:30], k["us_per_launch"]) for k in d.get("kernels") or [] if "shuffle" in k["kernel"]])'
for i in 1 2; do
echo "specialised shuffle convs" | tee -a $L; timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "generic shuffle convs" | tee -a $L; REFVSR_NO_CONV_SHUFFLE2=1 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
