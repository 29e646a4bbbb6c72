#!/bin/bash
# round-4 evidence run of the final tree (trimmed: the PMC passes, micro-benchmarks and the N = 2 protocol run were taken earlier
# today on unchanged kernels -- profiles/pmc_*.json, r04_*_microbench.txt, r04_bench_n2_gloo_one_gpu.json): full GPU suite, smoke,
# default bench, --no-cache, rocprofv3 kernel stats + trace analysis of the bench command, RefVSR_IR bench, two more fresh processes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_final2.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest -m gpu ==" | tee -a $L
timeout 700 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 240 --timeout-method=thread --durations=4 > gpurun_out/_t.out 2>&1
grep -i -A12 "Traceback\|^E " gpurun_out/_t.out | head -40 | cut -c1-300 | tee -a $L
tail -9 gpurun_out/_t.out | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r04_gpu_parity_report.txt 2>/dev/null
echo "== smoke ==" | tee -a $L
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $L
echo "== bench (default) ==" | tee -a $L
timeout 500 python bench.py --steps 20 --warmup 5 2> gpurun_out/r04_bench.err | tail -1 > gpurun_out/r04_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench.json'))
print('value', d['value'], d['samples'], 'min/median', d['min'] / d['median'])
print('dropin', d['dropin_surface']['value'], d['dropin_surface']['samples'])
print('roofline', {k: d['roofline'][k] for k in ('kernel','achieved','frac','mean_launch_ms','traffic')}, 'match', d['roofline_match_top2']['frac'])
print('whole_path', d['whole_path']['frac_of_f16_mfma_peak'], 'first_frame_ms', d['first_frame_ms'])
print('streams', json.dumps(d['streams']['median_pass']))
print('other', {k: (v.get('value'), v.get('samples'), v.get('roofline', {}).get('frac')) for k, v in d.get('other_configs', {}).items()})
print('cpu', d['cpu_baseline'])
w=d.get('wavefront_model', {}); print('phases', w.get('phase_ms_per_frame_measured')); print('wf8', json.dumps(w.get('predicted_speedup', {}).get('8'))[:1400])
for k in d.get('kernels', []): print('  %-70s %8.2f us  %6.1f TF (%.3f)  %7.1f GB/s (%.3f)' % (k['kernel'][:70], k['us_per_launch'], k['tflops'], k['frac_mfma'], k['gbs'], k['frac_hbm']))
" 2>&1 | cut -c1-1700 | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", d.get("samples"), "dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2))'
echo "== bench --no-cache (the reference's exact per-call work) ==" | tee -a $L
timeout 200 python bench.py --steps 20 --warmup 5 --repeats 3 --no-cache --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs 2>/dev/null | tail -1 > gpurun_out/r04_bench_nocache.json
python -c "import json; d=json.load(open('gpurun_out/r04_bench_nocache.json')); print('nocache value', d['value'], d['samples'])" | tee -a $L
echo "== rocprof ==" | tee -a $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200 | tee -a $L
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 > gpurun_out/r04_trace_analysis.txt 2>&1
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r04_trace_by_shape.txt 2>&1
head -22 gpurun_out/r04_trace_analysis.txt | tee -a $L
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r04_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof
echo "== RefVSR_IR_MFID (C = 36, EDVR refill; sequential engine) ==" | tee -a $L
timeout 200 python bench.py --config config_RefVSR_IR_MFID --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin --no-other-configs 2>/dev/null | tail -1 > gpurun_out/r04_bench_IR_MFID.json
python -c "import json; d=json.load(open('gpurun_out/r04_bench_IR_MFID.json')); print('IR_MFID', round(d['value'],2), 'fps', d['samples'])" 2>&1 | tail -1 | tee -a $L
echo "== cross-process stability of the headline: two more fresh processes ==" | tee -a $L
for i in 1 2; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs 2>/dev/null | tail -1 | python -c "$fmt" | tee -a $L
done
