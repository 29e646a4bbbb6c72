#!/bin/bash
# round 4, call 3: after the fixes of call 2 (batched SPyNet: two flows per resize launch; refvsr_warp_nhwc16_up2: shared
# align_corners taps with contraction off).  Every pytest is bounded per test (pytest-timeout) and in total.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call3.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== new op tests ==" | tee -a $L
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider --timeout 120 -k "conf_alpha or warp_up2 or batched_conv or store_modes" 2>&1 | tail -12 | tee -a $L
echo "== engine equivalence / pipelined / sharding ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 240 -x -k "round4 or pipelined or two_process or two_phase" 2>&1 | tail -12 | tee -a $L
echo "== full suite ==" | tee -a $L
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 240 --durations=5 2>&1 | tail -25 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r04_gpu_parity_report.txt 2>/dev/null
fmt='import sys,json
d=json.loads(sys.stdin.read())
print("value %.1f  samples %s  dropin %.1f  M %.2f P %.2f F %.2f ms/call" % (d["value"], d["samples"], d["dropin_surface"]["value"], d["streams"]["median_pass"]["M_ms_per_call"], d["streams"]["median_pass"]["P_ms_per_call"], d["streams"]["median_pass"]["F_ms_per_call"]))'
B="python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs"
run() {  # name, env assignments...
  local name=$1; shift
  echo "== $name ==" | tee -a $L
  env "$@" timeout 200 $B > gpurun_out/_b.out 2> gpurun_out/_b.err
  tail -1 gpurun_out/_b.out | python -c "$fmt" 2>/dev/null | cut -c1-300 | tee -a $L || true
  if ! tail -1 gpurun_out/_b.out | grep -q '"value"'; then tail -4 gpurun_out/_b.err | cut -c1-400 | tee -a $L; fi
}
for round in 1 2; do
  run "default (round $round)" X=1
  run "REFVSR_NO_SPYNET_BATCH=1 (round $round)" REFVSR_NO_SPYNET_BATCH=1
  run "REFVSR_RB24_STORE=1 (round $round)" REFVSR_RB24_STORE=1
  run "REFVSR_RB24_STORE=2 (round $round)" REFVSR_RB24_STORE=2
  run "round-3 launch list: NO_FUSE_CONF NO_SPYNET_BATCH NO_WARP_UP2 (round $round)" REFVSR_NO_FUSE_CONF=1 REFVSR_NO_SPYNET_BATCH=1 REFVSR_NO_WARP_UP2=1
  run "REFVSR_PIPE_DEPTH=2 (round $round)" REFVSR_PIPE_DEPTH=2
  run "REFVSR_BW_HEAD_BLOCKS=0 (round $round)" REFVSR_BW_HEAD_BLOCKS=0
  run "REFVSR_BW_HEAD_BLOCKS=12 (round $round)" REFVSR_BW_HEAD_BLOCKS=12
  run "REFVSR_BW_HEAD_BLOCKS=24 (round $round)" REFVSR_BW_HEAD_BLOCKS=24
done
echo "== resblock24 microbench (store modes) ==" | tee -a $L
RB_ITERS=6 timeout 200 python tools/bench_resblock.py 2>&1 | grep resblock | tee gpurun_out/r04_resblock_microbench.txt | grep -v "4 waves\|lean" | tee -a $L
echo "== spynet microbench ==" | tee -a $L
timeout 120 python tools/bench_spynet.py 2>&1 | grep "spynet" | tee gpurun_out/r04_spynet_microbench.txt | tee -a $L
echo "== rocprof kernel trace of the default bench ==" | tee -a $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200 | tee -a $L
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 > gpurun_out/r04_call3_trace_analysis.txt 2>&1
head -34 gpurun_out/r04_call3_trace_analysis.txt | tee -a $L
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r04_call3_trace_by_shape.txt 2>&1
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r04_call3_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof
echo "== full default bench ==" | tee -a $L
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_call3_bench.out 2> gpurun_out/r4_call3_bench.err
tail -1 gpurun_out/r4_call3_bench.out > gpurun_out/r4_call3_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r4_call3_bench.json'))
print('value', d['value'], d['samples'], 'dropin', d['dropin_surface']['value'], d['dropin_surface']['samples'], 'roofline', d['roofline']['frac'], d['roofline']['mean_launch_ms'])
print('other', {k: (v.get('value'), v.get('roofline', {}).get('frac')) for k, v in d.get('other_configs', {}).items()})
print('wavefront_model', json.dumps(d.get('wavefront_model'))[:1800])" 2>&1 | tee -a $L
tail -3 gpurun_out/r4_call3_bench.err | cut -c1-300 | tee -a $L
