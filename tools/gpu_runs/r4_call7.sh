#!/bin/bash
# round 4, call 7: (a) refvsr_conv_hr_last (conv_hr + head in one launch): op tests, engine tests, A/B in the frame; (b) the fused
# 48-channel block against two conv48 launches at 540 x 960 (the size rule's upper end); (c) kernel trace of a RefVSR_MFID frame
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call7.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== op tests ==" | tee -a $L
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider --timeout 100 --timeout-method=thread -k "conv_hr_last or conv_last or input_conv_8_plus_48 or resblock24" 2>&1 | tail -10 | tee -a $L
grep "conv_hr_last\|conv48 8+48" gpurun_out/gpu_ops_report.txt | tee -a $L
echo "== engine tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 240 --timeout-method=thread -k "round4 or stream_against_reference or full_size or x2 or pipelined" 2>&1 | tail -6 | tee -a $L
fmt='import sys,json
d=json.loads(sys.stdin.read())
print("value %.1f  samples %s  dropin %s  M %.2f P %.2f F %.2f" % (d["value"], d["samples"], d["dropin_surface"] and round(d["dropin_surface"]["value"],1), d["streams"]["median_pass"]["M_ms_per_call"], d["streams"]["median_pass"]["P_ms_per_call"], d["streams"]["median_pass"]["F_ms_per_call"]))'
B="python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs"
run() {
  local name=$1; shift
  echo "== $name ==" | tee -a $L
  env "$@" timeout 240 $B > gpurun_out/_b.out 2> gpurun_out/_b.err
  tail -1 gpurun_out/_b.out | python -c "$fmt" 2>/dev/null | cut -c1-300 | tee -a $L || true
  if ! tail -1 gpurun_out/_b.out | grep -q '"value"'; then tail -4 gpurun_out/_b.err | cut -c1-400 | tee -a $L; fi
}
for round in 1 2; do
  run "default (round $round)" X=1
  run "REFVSR_NO_FUSE_TAIL=1 (round $round)" REFVSR_NO_FUSE_TAIL=1   # (at the time of this run the fused tail was the default; it is opt-in since: REFVSR_FUSE_TAIL=1)
  run "default, REFVSR_BW_HEAD_BLOCKS=10 (round $round)" REFVSR_BW_HEAD_BLOCKS=10
done
echo "== tail microbench at 1080x1920: conv_hr + conv_last vs refvsr_conv_hr_last ==" | tee -a $L
timeout 120 python - <<'PY' 2>&1 | grep "^tail" | tee -a $L
import torch, sys
sys.path.insert(0, '.')
from refvsr_amd import ops
from refvsr_amd.packing import pack_conv, pack_conv_last, pack_conv_hr_last
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
w1, b1 = torch.randn(24, 24, 3, 3, generator=g) / 15, torch.randn(24, generator=g) * 0.1
w2, b2 = torch.randn(3, 24, 3, 3, generator=g) * 0.04, torch.randn(3, generator=g) * 0.1
x = ops.pack_nhwc16(torch.randn(24, 1080, 1920, generator=g).to(dev))
base = torch.rand(3, 270, 480, generator=g).to(dev)
cw = ops.ConvWeights(pack_conv(w1, b1, [24]), dev)
bl, bt = pack_conv_last(w2, b2).to(dev), pack_conv_hr_last(w1, b1, w2, b2).to(dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
a = t(lambda: ops.conv_last(bl, ops.conv(cw, x, act=0.1), base))
b = t(lambda: ops.conv_hr_last(bt, x, base))
print('tail 1080x1920: conv_hr + conv_last (two launches) %.1f us | refvsr_conv_hr_last %.1f us' % (a, b))
PY
echo "== resblock48 at 540x960: fused vs two conv48 launches ==" | tee -a $L
timeout 120 python - <<'PY' 2>&1 | grep "^rb48" | tee -a $L
import torch, sys
sys.path.insert(0, '.')
from refvsr_amd import ops
from refvsr_amd.packing import pack_conv
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
C, n = 48, 6
pairs = [tuple(ops.ConvWeights(pack_conv(torch.randn(C, C, 3, 3, generator=g) / 30, torch.randn(C, generator=g) * 0.01, [C]), dev) for _ in range(2)) for _ in range(n)]
ch = ops.Resblock48Chain(pairs, dev)
def two(x, act):
    for c1, c2 in pairs:
        x = ops.conv(c2, ops.conv(c1, x, act=act), res=x)
    return x
def t(fn, it):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    big = torch.randn(8192, 8192, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _ = big @ big
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3 / n
for h, w in ((540, 960), (405, 720)):
    x = ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev))
    print('rb48 %dx%d: two conv48 launches %.1f us/block | fused %.1f us/block | equal %s' % (h, w, t(lambda: two(x, 0.2), 6), t(lambda: ops.resblock48_chain(ch, x, 0.2), 6), bool(torch.equal(two(x, 0.2), ops.resblock48_chain(ch, x, 0.2)))))
PY
echo "== kernel trace of RefVSR_MFID ==" | tee -a $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --config config_RefVSR_MFID --steps 12 --warmup 3 --repeats 1 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r04_trace_by_shape_MFID.txt 2>&1
head -24 gpurun_out/r04_trace_by_shape_MFID.txt | cut -c1-175 | tee -a $L
rm -rf gpurun_out/prof
