#!/bin/bash
# round 4, call 6: the two-source 48 + 48 -> 48 conv as two channel halves (refvsr_conv48, NCG = 12 plan): op tests, MFID engine
# tests, microbench, MFID / 8K frame rates with and without it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call6.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== op tests ==" | tee -a $L
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider --timeout 100 --timeout-method=thread -k "two_source_channel_halves or conv24_specialised or conv_shuffle2 or conv32 or conf_alpha or conv_last" 2>&1 | tail -12 | tee -a $L
grep "conv48 48+48" gpurun_out/gpu_ops_report.txt | tee -a $L
echo "== MFID / HD48 engine tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 240 --timeout-method=thread -k "MFID or HD48 or F_16x24 or round4 or 8k or mfid" 2>&1 | tail -8 | tee -a $L
echo "== microbench: 48 + 48 -> 48, channel halves vs the generic streamed kernel ==" | tee -a $L
timeout 200 python - <<'PY' 2>&1 | grep "conv48x2" | tee gpurun_out/r04_conv48x2_microbench.txt | tee -a $L
import torch, sys
sys.path.insert(0, '.')
from refvsr_amd import ops
from refvsr_amd.packing import pack_conv
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
wt, b = torch.randn(48, 96, 3, 3, generator=g) / 30, torch.randn(48, generator=g) * 0.1
cw = ops.ConvWeights(pack_conv(wt, b, [48, 48]), dev)
def t(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    big = torch.randn(8192, 8192, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _ = big @ big
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, h, w, n in (('LR 270x480', 270, 480, 30), ('2x 540x960', 540, 960, 20), ('1080x1920', 1080, 1920, 8), ('2160x3840', 2160, 3840, 3)):
    a_, b_ = ops.pack_nhwc16(torch.randn(48, h, w, generator=g).to(dev)), ops.pack_nhwc16(torch.randn(48, h, w, generator=g).to(dev))
    fl = 2.0 * 9 * 96 * 48 * h * w
    new = t(lambda: ops.conv(cw, a_, b_, act=0.2), n)
    blob, cw.blob24 = cw.blob24, None
    old = t(lambda: ops.conv(cw, a_, b_, act=0.2), n)
    cw.blob24 = blob
    print('conv48x2 %-12s channel halves %8.1f us = %6.1f TFLOP/s (%.1f %% of 2.5 PF) | generic streamed %8.1f us = %6.1f TFLOP/s' % (name, new, fl / new / 1e6, fl / new / 1e6 / 25.0, old, fl / old / 1e6), flush=True)
PY
fmt='import sys,json
d=json.loads(sys.stdin.read())
print("value %.2f  samples %s  dropin %s" % (d["value"], d["samples"], d["dropin_surface"] and round(d["dropin_surface"]["value"],1)))'
run() {
  local name=$1; shift
  echo "== $name ==" | tee -a $L
  env "$@" timeout 300 python bench.py --repeats 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs $EXTRA > gpurun_out/_b.out 2> gpurun_out/_b.err
  tail -1 gpurun_out/_b.out | python -c "$fmt" 2>/dev/null | cut -c1-300 | tee -a $L || true
  if ! tail -1 gpurun_out/_b.out | grep -q '"value"'; then tail -4 gpurun_out/_b.err | cut -c1-400 | tee -a $L; fi
}
EXTRA="--config config_RefVSR_MFID --steps 12 --warmup 3"
for round in 1 2; do
  run "RefVSR_MFID default (round $round)" X=1
  run "RefVSR_MFID REFVSR_NO_CONV48X2=1 (round $round)" REFVSR_NO_CONV48X2=1
done
echo "== RefVSR_MFID_8K 1080p -> 8K with and without ==" | tee -a $L
for K in X REFVSR_NO_CONV48X2; do
  env $K=1 timeout 300 python -c "
import bench, torch, json
r = bench.other_config_leg('config_RefVSR_MFID_8K', 1080, 1920, 4, 2, torch.device('cuda:0'), repeats=2)
print('$K', round(r['value'], 3), r['samples'], r['peak_memory_gib'])" 2>&1 | tail -1 | tee -a $L
done
