#!/bin/bash
# round-4 evidence run: full GPU suite, smoke, PMC passes (-> profiles/pmc_*.json), default bench (+ --no-cache, --no-pipeline),
# rocprofv3 kernel stats + trace analysis of the bench command, RefVSR_IR bench, N = 2 protocol over gloo on one GPU, micro-benchmarks.
# Every step is bounded (pytest-timeout per test, `timeout` per step).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_final.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest -m gpu ==" | tee -a $L
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 240 --timeout-method=thread --durations=4 2>&1 | tail -12 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r04_gpu_parity_report.txt 2>/dev/null
echo "== smoke ==" | tee -a $L
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $L
echo "== pmc ==" | tee -a $L
rm -rf gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_fetch" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_write" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_write.log" 2>&1)
python tools/pmc_to_json.py gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write profiles 2>&1 | tail -3 | tee -a $L
cp profiles/pmc_kernels.json profiles/pmc_match_top2.json gpurun_out/ 2>/dev/null
find gpurun_out/pmc_k_fetch -name "*counter_collection.csv" -exec cp {} gpurun_out/r04_pmc_kernels_FETCH_SIZE.csv \;
find gpurun_out/pmc_k_write -name "*counter_collection.csv" -exec cp {} gpurun_out/r04_pmc_kernels_WRITE_SIZE.csv \;
rm -rf gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write
echo "== bench (default), shader clock / package power sampled once a second next to it ==" | tee -a $L
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Graphics Package" | tr '\n' ' '; echo; sleep 1; done > gpurun_out/r04_clocks_during_bench.txt ) &
SMI=$!
timeout 600 python bench.py --steps 20 --warmup 5 2> gpurun_out/r04_bench.err | tail -1 > gpurun_out/r04_bench.json
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r04_bench.json'))
print('value', d['value'], d['samples'], 'min/median', d['min'] / d['median'])
print('dropin', d['dropin_surface']['value'], d['dropin_surface']['samples'])
print('roofline', {k: d['roofline'][k] for k in ('kernel','achieved','frac','mean_launch_ms','traffic')}, 'match', d['roofline_match_top2']['frac'])
print('whole_path', d['whole_path']['frac_of_f16_mfma_peak'], 'first_frame_ms', d['first_frame_ms'])
print('streams', json.dumps(d['streams']['median_pass']))
print('other', {k: (v.get('value'), v.get('samples'), v.get('roofline', {}).get('frac')) for k, v in d.get('other_configs', {}).items()})
print('cpu', d['cpu_baseline'])
print('wf8', json.dumps(d.get('wavefront_model', {}).get('predicted_speedup', {}).get('8')))
for k in d.get('kernels', []): print('  %-70s %8.2f us  %6.1f TF (%.3f)  %7.1f GB/s (%.3f)' % (k['kernel'][:70], k['us_per_launch'], k['tflops'], k['frac_mfma'], k['gbs'], k['frac_hbm']))
" 2>&1 | cut -c1-1500 | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", d.get("samples"), "dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2))'
echo "== bench --no-pipeline (frame ids, one call at a time) ==" | tee -a $L
timeout 300 python bench.py --steps 20 --warmup 5 --no-pipeline --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs 2>/dev/null | tail -1 | python -c "$fmt" | tee -a $L
echo "== bench --no-cache (the reference's exact per-call work) ==" | tee -a $L
timeout 300 python bench.py --steps 20 --warmup 5 --no-cache --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs 2>/dev/null | tail -1 > gpurun_out/r04_bench_nocache.json
python -c "import json; d=json.load(open('gpurun_out/r04_bench_nocache.json')); print('nocache value', d['value'], d['samples'])" | tee -a $L
echo "== rocprof ==" | tee -a $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200 | tee -a $L
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 > gpurun_out/r04_trace_analysis.txt 2>&1
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r04_trace_by_shape.txt 2>&1
head -26 gpurun_out/r04_trace_analysis.txt | tee -a $L
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r04_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof
echo "== RefVSR_IR_MFID (C = 36, EDVR refill; sequential engine) ==" | tee -a $L
timeout 300 python bench.py --config config_RefVSR_IR_MFID --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin --no-other-configs 2>/dev/null | tail -1 > gpurun_out/r04_bench_IR_MFID.json
python -c "import json; d=json.load(open('gpurun_out/r04_bench_IR_MFID.json')); print('IR_MFID', round(d['value'],2), 'fps', d['samples'])" 2>&1 | tail -1 | tee -a $L
echo "== N = 2 protocol, two ranks on one GPU over gloo ==" | tee -a $L
REFVSR_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --repeats 2 --clip 20 --no-kernels --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-9000 > gpurun_out/r04_bench_n2_gloo_one_gpu.json
python -c "import json; d=json.load(open('gpurun_out/r04_bench_n2_gloo_one_gpu.json')); w=d.get('wavefront'); print('n2 value', d['value'], 'wavefront', {k: w.get(k) for k in ('ranks_seen','value','frames_equal','partition')} if w else None)" 2>&1 | cut -c1-700 | tee -a $L
echo "== micro-benchmarks ==" | tee -a $L
RB_ITERS=6 timeout 200 python tools/bench_resblock.py 2>&1 | grep resblock | tee gpurun_out/r04_resblock_microbench.txt | grep -v "4 waves\|lean\|sc1" | tee -a $L
timeout 120 python tools/bench_spynet.py 2>&1 | grep "spynet" | tee gpurun_out/r04_spynet_microbench.txt | tee -a $L
timeout 120 python tools/bench_match.py 2>&1 | grep match_top2 | tee gpurun_out/r04_match_microbench.txt | tee -a $L
echo "== cross-process stability of the headline: three more fresh processes ==" | tee -a $L
for i in 1 2 3; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs 2>/dev/null | tail -1 | python -c "$fmt" | tee -a $L
done
echo "== RefVSR_MFID: backward head on the preparation stream (-1 = off | 12 | 15) ==" | tee -a $L
for H in -1 12 15; do
  echo "REFVSR_BW_HEAD_BLOCKS=$H" | tee -a $L
  REFVSR_BW_HEAD_BLOCKS=$H timeout 240 python bench.py --config config_RefVSR_MFID --steps 12 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs 2>/dev/null | tail -1 | python -c "$fmt" | tee -a $L
done
