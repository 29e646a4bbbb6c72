#!/bin/bash
# round-5 evidence run: full GPU suite, smoke, default bench (the driver's command), rocprofv3 kernel stats + trace analysis of the
# bench command (group mode and one frame per call), --no-cache, RefVSR_IR bench, a second fresh bench process, --gpus 2 on this one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r05_final_run.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest -m gpu ==" | tee -a $L
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 600 --timeout-method=thread 2>&1 | tail -6 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r05_gpu_parity_report.txt 2>/dev/null
echo "== smoke ==" | tee -a $L
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $L
echo "== default bench (the driver's command) ==" | tee -a $L
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json gpurun_out/r05_bench_full.json > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
echo "rc $? line bytes $(wc -c < gpurun_out/r05_bench.json)" | tee -a $L
cat gpurun_out/r05_bench.json | tee -a $L
echo "== rocprofv3 kernel stats of the bench command ==" | tee -a $L
for g in 4 1; do
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --group $g --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
  f=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
  st=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  sfx=""; [ $g = 1 ] && sfx="_one_frame_per_call"
  cp $st gpurun_out/r05_bench_kernel_stats$sfx.csv
  python tools/trace_analysis.py $f 8 20 > gpurun_out/r05_trace_analysis$sfx.txt 2>&1
  python tools/trace_by_shape.py $f 300 > gpurun_out/r05_trace_by_shape$sfx.txt 2>&1
  echo "-- group $g --" | tee -a $L
  head -12 gpurun_out/r05_trace_analysis$sfx.txt | tee -a $L
  grep "resblock24_kernel" gpurun_out/r05_bench_kernel_stats$sfx.csv | head -6 | tee -a $L
  rm -rf gpurun_out/prof
done
fmt='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f  samples %s  percall %s dropin %s roofline %s" % (d["value"], d["samples"], d.get("one_frame_per_call") and d["one_frame_per_call"]["value"], d.get("dropin_surface") and d["dropin_surface"]["value"], d["roofline"] and (round(d["roofline"].get("frac"),4), d["roofline"].get("mean_launch_ms"))))'
echo "== second fresh process ==" | tee -a $L
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --full-json gpurun_out/_b.json | python -c "$fmt" | tee -a $L
echo "== --no-cache (what the reference executes) ==" | tee -a $L
timeout 300 python bench.py --steps 20 --warmup 5 --no-cache --repeats 3 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --full-json gpurun_out/_b.json | python -c "$fmt" | tee -a $L
echo "== RefVSR_IR_MFID ==" | tee -a $L
timeout 300 python bench.py --config config_RefVSR_IR_MFID --steps 12 --warmup 5 --repeats 3 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --full-json gpurun_out/_b.json | python -c "$fmt" | tee -a $L
echo "== --gpus 2 on this one GPU (gloo) ==" | tee -a $L
REFVSR_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 8 --warmup 5 --repeats 2 --no-kernels --full-json gpurun_out/r05_bench_n2_full.json > gpurun_out/r05_bench_n2_gloo_one_gpu.json 2> gpurun_out/_n2.err
echo "rc $? bytes $(wc -c < gpurun_out/r05_bench_n2_gloo_one_gpu.json)" | tee -a $L
tail -1 gpurun_out/r05_bench_n2_gloo_one_gpu.json | cut -c1-2500 | tee -a $L
