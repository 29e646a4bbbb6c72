#!/bin/bash
# round 4, call 5: the fused output head (refvsr_conv_last): op tests, engine test, A/B in the frame
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call5.log
: > $L
echo "== conv_last op tests ==" | tee -a $L
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider --timeout 100 --timeout-method=thread -k "conv_last" 2>&1 | tail -12 | tee -a $L
grep "conv_last" gpurun_out/gpu_ops_report.txt | tail -16 | tee -a $L
echo "== engine: equivalence test, fixtures, full-size ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 240 --timeout-method=thread -k "round4 or stream_against_reference or full_size or x2" 2>&1 | tail -8 | tee -a $L
fmt='import sys,json
d=json.loads(sys.stdin.read())
print("value %.1f  samples %s  dropin %s  M %.2f P %.2f F %.2f" % (d["value"], d["samples"], d["dropin_surface"] and round(d["dropin_surface"]["value"],1), d["streams"]["median_pass"]["M_ms_per_call"], d["streams"]["median_pass"]["P_ms_per_call"], d["streams"]["median_pass"]["F_ms_per_call"]))'
B="python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs"
run() {
  local name=$1; shift
  echo "== $name ==" | tee -a $L
  env "$@" timeout 240 $B $EXTRA > gpurun_out/_b.out 2> gpurun_out/_b.err
  tail -1 gpurun_out/_b.out | python -c "$fmt" 2>/dev/null | cut -c1-300 | tee -a $L || true
  if ! tail -1 gpurun_out/_b.out | grep -q '"value"'; then tail -4 gpurun_out/_b.err | cut -c1-400 | tee -a $L; fi
}
EXTRA=""
for round in 1 2; do
  run "default (round $round)" X=1
  run "REFVSR_NO_FUSE_HEAD=1 (round $round)" REFVSR_NO_FUSE_HEAD=1
  run "default, REFVSR_BW_HEAD_BLOCKS=14 (round $round)" REFVSR_BW_HEAD_BLOCKS=14
done
echo "== head microbench: fused vs bicubic + generic conv at 1080x1920 ==" | tee -a $L
timeout 120 python - <<'PY' 2>&1 | tail -4 | tee -a $L
import torch, sys
sys.path.insert(0, '.')
from refvsr_amd import ops
from refvsr_amd.packing import pack_conv, pack_conv_last
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for c in (24, 48):
    w_, b_ = torch.randn(3, c, 3, 3, generator=g) * 0.03, torch.randn(3, generator=g) * 0.1
    x = ops.pack_nhwc16(torch.randn(c, 1080, 1920, generator=g).to(dev))
    base = torch.rand(3, 270, 480, generator=g).to(dev)
    cw = ops.ConvWeights(pack_conv(w_, b_, [c]), dev)
    blob = pack_conv_last(w_, b_).to(dev)
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    a = t(lambda: ops.conv(cw, x, planar_out=True, res_planar=ops.bicubic_scale(base, 4, clamp01=True), clamp=(0.0, 1.0)))
    b = t(lambda: ops.conv_last(blob, x, base))
    byts = 1080 * 1920 * (c * 2 + 12) + 3 * 270 * 480 * 4
    print('head C=%d 1080x1920: bicubic + generic planar conv %.1f us | refvsr_conv_last %.1f us = %.0f GB/s of %d MB algorithmic' % (c, a, b, byts / b / 1e3, byts // 10 ** 6))
PY
echo "== the default bench again (fused head in), no rocm-smi loop next to it: the evidence line of the round ==" | tee -a $L
timeout 600 python bench.py --steps 20 --warmup 5 2> gpurun_out/r04_bench.err | tail -1 > gpurun_out/r04_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench.json'))
print('value', d['value'], d['samples'], 'min/median', d['min'] / d['median'])
print('dropin', d['dropin_surface']['value'], d['dropin_surface']['samples'])
print('roofline', {k: d['roofline'][k] for k in ('achieved','frac','mean_launch_ms','traffic')}, 'match', d['roofline_match_top2']['frac'])
print('whole_path', d['whole_path']['frac_of_f16_mfma_peak'], 'first_frame_ms', d['first_frame_ms'])
print('streams', json.dumps(d['streams']['median_pass']))
print('other', {k: (v.get('value'), v.get('samples'), v.get('roofline', {}).get('frac')) for k, v in d.get('other_configs', {}).items()})
print('cpu', d['cpu_baseline']['seconds_per_frame'], d['cpu_baseline']['cores'])
print('wf', json.dumps(d.get('wavefront_model', {}).get('phase_ms_per_frame_measured')), json.dumps(d.get('wavefront_model', {}).get('predicted_speedup', {}).get('8')))
" 2>&1 | cut -c1-1200 | tee -a $L
echo "== rocprof of the default bench ==" | tee -a $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 > gpurun_out/r04_trace_analysis.txt 2>&1
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r04_trace_by_shape.txt 2>&1
head -12 gpurun_out/r04_trace_analysis.txt | tee -a $L
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r04_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof
echo "== full suite (fused head in) ==" | tee -a $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 240 --timeout-method=thread 2>&1 | tail -6 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r04_gpu_parity_report.txt 2>/dev/null
