#!/bin/bash
# round-3 evidence run: full GPU suite, smoke, PMC passes (-> profiles/pmc_*.json), default bench (+ --no-cache, --no-pipeline),
# rocprofv3 kernel stats + trace analysis of the bench command, MFID / MFID_8K benches, N = 2 protocol over gloo on one GPU,
# resblock24 probe + micro-benchmark
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3_final.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest -m gpu ==" | tee -a $L
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=4 2>&1 | tail -12 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r03_gpu_parity_report.txt 2>/dev/null
echo "== smoke ==" | tee -a $L
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $L
echo "== pmc ==" | tee -a $L
rm -rf gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_fetch" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_write" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_write.log" 2>&1)
python tools/pmc_to_json.py gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write profiles 2>&1 | tail -3 | tee -a $L
cp profiles/pmc_kernels.json profiles/pmc_match_top2.json gpurun_out/ 2>/dev/null
find gpurun_out/pmc_k_fetch -name "*counter_collection.csv" -exec cp {} gpurun_out/r03_pmc_kernels_FETCH_SIZE.csv \;
find gpurun_out/pmc_k_write -name "*counter_collection.csv" -exec cp {} gpurun_out/r03_pmc_kernels_WRITE_SIZE.csv \;
rm -rf gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write
echo "== SQ counters of the resblock kernels ==" | tee -a $L
for pass in 1 2; do
  case $pass in
    1) CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES";;
    2) CTRS="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE";;
  esac
  rm -rf gpurun_out/pmc_rb$pass
  (cd /tmp && timeout 200 rocprofv3 --pmc $CTRS --output-format csv -d "$OLDPWD/gpurun_out/pmc_rb$pass" -o k -- python "$OLDPWD/tools/pmc_resblock.py" > "$OLDPWD/gpurun_out/pmc_rb$pass.log" 2>&1)
  python tools/pmc_summary.py gpurun_out/pmc_rb$pass resblock > gpurun_out/r03_pmc_sq_resblock_pass$pass.txt
  rm -rf gpurun_out/pmc_rb$pass
done
echo "== bench (default), shader clock / package power sampled once a second next to it ==" | tee -a $L
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Graphics Package" | tr '\n' ' '; echo; sleep 1; done > gpurun_out/r03_clocks_during_bench.txt ) &
SMI=$!
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r03_bench.json
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
python - <<'PY' | tee -a $L
import re
rows = []
for l in open('gpurun_out/r03_clocks_during_bench.txt'):
    m, p = re.search(r'\((\d+)Mhz\)', l), re.search(r'Power \(W\): ([0-9.]+)', l)
    if m and p:
        rows.append((int(m.group(1)), float(p.group(1))))
busy = [r for r in rows if r[1] > 350]
if busy:
    c = sorted(r[0] for r in busy); w = sorted(r[1] for r in busy)
    print('clock samples under load: %d of %d; sclk min/median/max %d/%d/%d MHz; power median/max %.0f/%.0f W' % (len(busy), len(rows), c[0], c[len(c) // 2], c[-1], w[len(w) // 2], w[-1]))
else:
    print('no loaded clock samples (%d rows)' % len(rows))
PY
python -c "import json; d=json.load(open('gpurun_out/r03_bench.json')); print('value', d['value'], 'dropin', d['dropin_surface']['value'], 'roofline', {k: d['roofline'][k] for k in ('kernel','achieved','frac','mean_launch_ms','traffic')}, 'match', d['roofline_match_top2']['frac'], 'cpu', d['cpu_baseline'], 'wf', d.get('wavefront_model', {}).get('predicted_speedup'))" | cut -c1-1200 | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2))'
echo "== bench --no-pipeline (frame ids, one call at a time) ==" | tee -a $L
timeout 300 python bench.py --steps 40 --warmup 3 --no-pipeline --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "== bench --no-cache ==" | tee -a $L
timeout 300 python bench.py --no-cache --no-cpu-baseline --no-kernels --no-dropin --no-wavefront 2>&1 | tail -1 > gpurun_out/r03_bench_nocache.json
python -c "import json; d=json.load(open('gpurun_out/r03_bench_nocache.json')); print('nocache value', d['value'])" | tee -a $L
echo "== rocprof ==" | tee -a $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline --no-kernels --no-dropin --no-wavefront > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200 | tee -a $L
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 > gpurun_out/r03_trace_analysis.txt 2>&1
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r03_trace_by_shape.txt 2>&1
head -26 gpurun_out/r03_trace_analysis.txt | tee -a $L
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r03_bench_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof/bench_domain_stats.csv gpurun_out/r03_bench_domain_stats.csv 2>/dev/null
rm -rf gpurun_out/prof
echo "== MFID (configs[2]) ==" | tee -a $L
timeout 400 python bench.py --config config_RefVSR_MFID --steps 20 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 > gpurun_out/r03_bench_MFID.json
python -c "$fmt" < gpurun_out/r03_bench_MFID.json | tee -a $L
echo "== RefVSR_IR_MFID (C = 36, EDVR refill; sequential engine) ==" | tee -a $L
timeout 400 python bench.py --config config_RefVSR_IR_MFID --steps 10 --warmup 2 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin 2>&1 | tail -1 > gpurun_out/r03_bench_IR_MFID.json
python -c "import json; d=json.load(open('gpurun_out/r03_bench_IR_MFID.json')); print('IR_MFID', round(d['value'],2), 'fps', round(d['ms_per_step'],2), 'ms')" 2>&1 | tail -1 | tee -a $L
echo "== MFID_8K 1080p (configs[4]) ==" | tee -a $L
timeout 600 python bench.py --config config_RefVSR_MFID_8K --size 1080x1920 --frames 5 --steps 6 --warmup 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront 2>&1 | tail -1 > gpurun_out/r03_bench_MFID_8K_1080p.json
python -c "import json; d=json.load(open('gpurun_out/r03_bench_MFID_8K_1080p.json')); print('8K', round(d['value'],2), 'fps', round(d['ms_per_step'],1), 'ms')" | tee -a $L
echo "== N = 2 protocol, two ranks on one GPU over gloo ==" | tee -a $L
REFVSR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --clip 20 --no-kernels --no-cpu-baseline 2>&1 | tail -1 | cut -c1-6000 > gpurun_out/r03_bench_n2_gloo_one_gpu.json
python -c "import json; d=json.load(open('gpurun_out/r03_bench_n2_gloo_one_gpu.json')); w=d.get('wavefront'); print('n2 value', d['value'], 'wavefront', {k: w.get(k) for k in ('value','frames_equal','partition','phase_ms_per_frame_measured','handoff')} if w else None)" 2>&1 | cut -c1-900 | tee -a $L
echo "== resblock24 probe + micro-benchmark ==" | tee -a $L
timeout 300 python tools/probe_resblock24.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_probe_resblock24.txt
timeout 300 python tools/bench_resblock.py 2>&1 | grep resblock | tee gpurun_out/r03_resblock_microbench.txt | grep -v "4 waves" | tee -a $L
timeout 200 python tools/bench_spynet.py 2>&1 | grep "spynet flow" | tee gpurun_out/r03_spynet_microbench.txt | tee -a $L
REFVSR_SPYNET_HILO=1 timeout 200 python tools/bench_spynet.py 2>&1 | grep "spynet flow" | tee -a gpurun_out/r03_spynet_microbench.txt | tee -a $L
timeout 120 python tools/bench_match.py 2>&1 | grep match_top2 | tee gpurun_out/r03_match_microbench.txt | tee -a $L
