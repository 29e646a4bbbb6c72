#!/bin/bash
# round 6, final tree: the driver's three commands in the driver's form, the sustained-load bench, the N = 2 protocol run of configs[3]
# at its size, rocprofv3 kernel stats + trace analysis of the bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_final_tree_run.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== python -m pytest tests/ -x -q -m gpu ==" | tee -a $L
( time timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) 2>&1 | tail -8 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r06_gpu_parity_report.txt 2>/dev/null
cp /tmp/bench_n2_configs3_full.json gpurun_out/r06_bench_n2_configs3_gloo_one_gpu.json 2>/dev/null
echo "== smoke ==" | tee -a $L
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $L
echo "== python bench.py --gpus 1 --steps 20 --warmup 5 ==" | tee -a $L
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json gpurun_out/r06_bench_full.json > gpurun_out/r06_bench.json 2> gpurun_out/_b.err ) 2>&1 | grep real | tee -a $L
cat gpurun_out/r06_bench.json | tee -a $L
echo "== the same box, 100 timed steps per pass (sustained load) ==" | tee -a $L
timeout 600 python bench.py --steps 100 --warmup 5 --repeats 3 --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront \
    --full-json gpurun_out/_s100_full.json > gpurun_out/r06_bench_100_steps.json 2> gpurun_out/_s100.err
python - <<'PY' | tee -a $L
import json
j=json.load(open('gpurun_out/r06_bench_100_steps.json'))
print('100 steps: groups', round(j['value'],2), j['samples'], 'per-call', j['one_frame_per_call']['value'], 'dropin', j['dropin_surface']['value'], 'pcie', j['pcie_inclusive'], 'frac', j['roofline']['frac'] if j.get('roofline') else None)
PY
echo "== rocprofv3 --kernel-trace --stats of the bench command ==" | tee -a $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
f=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
st=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
cp $st gpurun_out/r06_bench_kernel_stats.csv
python tools/trace_analysis.py $f 8 20 > gpurun_out/r06_trace_analysis.txt 2>&1
python tools/trace_by_shape.py $f 300 > gpurun_out/r06_trace_by_shape.txt 2>&1
head -14 gpurun_out/r06_trace_analysis.txt | tee -a $L
head -12 gpurun_out/r06_bench_kernel_stats.csv | tee -a $L
rm -rf gpurun_out/prof
