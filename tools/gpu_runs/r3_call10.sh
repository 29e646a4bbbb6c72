#!/bin/bash
# shader clock under the matching GEMM and under the bench (is the 45-50 % of the nominal peak a power / clock limit?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call10.log
: > $L
rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk" | head -4 | tee -a $L
sample() { for i in $(seq 1 $1); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)\|Graphics Package" | tr '\n' ' '; echo; sleep 0.5; done; }
echo "== idle ==" | tee -a $L
sample 2 | tee -a $L
echo "== match_top2 loop (5 s) ==" | tee -a $L
( timeout 60 python - <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from refvsr_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
lr_f = torch.randn(16, 270, 480, generator=g).to(dev); ref_f = torch.randn(16, 270, 480, generator=g).to(dev)
lr_rows, _ = ops.match_patches(lr_f, 512); ref_rows, _ = ops.match_patches(ref_f, 256)
n = 270 * 480
t0 = time.time()
while time.time() - t0 < 8:
    for _ in range(20): ops.match_top2(ref_rows, n, lr_rows, n, 1)
    torch.cuda.synchronize()
PY
) &
sleep 4
sample 6 | tee -a $L
wait
echo "== bench loop ==" | tee -a $L
( timeout 200 python bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin > gpurun_out/r3_call10_bench.json 2>&1 ) &
sleep 25
sample 6 | tee -a $L
wait
tail -1 gpurun_out/r3_call10_bench.json | cut -c1-200 | tee -a $L
