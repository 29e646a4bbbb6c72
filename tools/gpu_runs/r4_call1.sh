#!/bin/bash
# round 4, call 1: the pipelined mode's stream layout.  (a) the new bit-identity test over every layout /
# input_ready form, (b) layout x hardware-queue-count A/B through bench.py's repeated, interleaved passes (GPU_MAX_HW_QUEUES=8
# reproduced round 3's slow mode: every stream on its own queue), (c) the full default bench line (other_configs included),
# (d) kernel trace of the default bench for the launch count / per-kernel times the next steps start from
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call1.log
: > $L
echo "== tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x -k "pipelined or streams_are or two_phase or static_input or batch_and_api" 2>&1 | tail -5 | tee -a $L
fmt='import sys,json
d=json.loads(sys.stdin.read())
print("value %.1f  samples %s  dropin %.1f %s  streams %s" % (d["value"], d["samples"], d["dropin_surface"]["value"], d["dropin_surface"]["samples"], json.dumps(d["streams"]["median_pass"])))
if d["streams"]["slow_passes"]: print("   SLOW", json.dumps(d["streams"]["slow_passes"]))'
for Q in default 8 2; do
  for LAY in pf_m pfm p_fm; do
    echo "== layout $LAY, GPU_MAX_HW_QUEUES=$Q ==" | tee -a $L
    if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
    REFVSR_PIPE_LAYOUT=$LAY timeout 300 python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs 2>&1 | tail -1 > gpurun_out/r4_layout_${LAY}_q${Q}.json
    python -c "$fmt" < gpurun_out/r4_layout_${LAY}_q${Q}.json 2>&1 | cut -c1-700 | tee -a $L
  done
done
unset GPU_MAX_HW_QUEUES
echo "== second round of the default layout (fresh processes: is the rate bimodal across processes?) ==" | tee -a $L
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs 2>&1 | tail -1 | python -c "$fmt" 2>&1 | cut -c1-400 | tee -a $L
  REFVSR_PIPE_LAYOUT=pfm timeout 300 python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs 2>&1 | tail -1 | python -c "$fmt" 2>&1 | cut -c1-400 | tee -a $L
done
echo "== full default bench ==" | tee -a $L
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_call1_bench.out 2> gpurun_out/r4_call1_bench.err
tail -1 gpurun_out/r4_call1_bench.out > gpurun_out/r4_call1_bench.json
tail -3 gpurun_out/r4_call1_bench.err | cut -c1-300 | tee -a $L
python -c "
import json; d=json.load(open('gpurun_out/r4_call1_bench.json'))
print('value', d['value'], d['samples'], 'dropin', d['dropin_surface']['value'], 'roofline', d['roofline']['frac'], d['roofline']['mean_launch_ms'])
print('other', json.dumps(d.get('other_configs'))[:1500])
print('cpu', d.get('cpu_baseline'))" 2>&1 | tee -a $L
echo "== rocprof kernel trace of the default bench ==" | tee -a $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200 | tee -a $L
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 > gpurun_out/r04_call1_trace_analysis.txt 2>&1
head -30 gpurun_out/r04_call1_trace_analysis.txt | tee -a $L
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r04_call1_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof
