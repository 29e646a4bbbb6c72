#!/bin/bash
# round 5, call 19: match_patches through LDS + match_refine's two-candidate pass against the previous build, bit for bit; matching tests; bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r05_call19.log
: > $L
echo "== A/B against the previous build ==" | tee -a $L
timeout 600 python tools/ab_lib_compare.py 2>&1 | grep -v amdgpu.ids | tee -a $L
echo "== matching tests ==" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q -m gpu -p no:cacheprovider -k "match or full_size or batch_samples" 2>&1 | tail -5 | tee -a $L
echo "== default bench ==" | tee -a $L
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs > gpurun_out/r05_bench_call19.json 2> gpurun_out/_b.err ) 2>&1 | grep real | tee -a $L
tail -2 gpurun_out/_b.err | tee -a $L
python - <<'PY' | tee -a $L
import json
j=json.load(open('gpurun_out/r05_bench_call19.json'))
print('value', j['value'], j['samples'], 'percall', j['one_frame_per_call']['value'], 'dropin', j['dropin_surface']['value'], 'pcie', j['pcie_inclusive'], 'frac', j['roofline']['frac'])
PY
