#!/bin/bash
# round 5, call 22: the strided predictor conv in 4 x 32 tile mode (16 output channels per workgroup) instead of gather mode
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r05_pconv_mt1_ab.txt
: > $L
timeout 300 python tools/bench_pconv.py 2>&1 | grep -v amdgpu.ids | tee -a $L
for rep in 1 2; do
for mt in 2 1; do
  REFVSR_PCONV_MT=$mt timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront --repeats 3 \
      --full-json gpurun_out/_pc_full.json > gpurun_out/_pc.json 2> gpurun_out/_pc.err
  python - <<PY | tee -a $L
import json
j=json.load(open('gpurun_out/_pc.json'))
print('REFVSR_PCONV_MT=$mt rep $rep: value', round(j['value'],2), j['samples'], 'per-call', j['one_frame_per_call']['value'], 'dropin', j['dropin_surface']['value'])
PY
done
done
