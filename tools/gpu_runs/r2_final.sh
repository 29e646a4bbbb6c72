#!/bin/bash
# round-2 final evidence: full GPU suite, smoke, PMC passes, default bench (+ --no-cache), rocprofv3 kernel stats + trace analysis
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_final.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest -m gpu ==" | tee -a $L
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=4 2>&1 | tail -12 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r02_gpu_parity_report.txt 2>/dev/null
echo "== smoke ==" | tee -a $L
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $L
echo "== pmc ==" | tee -a $L
rm -rf gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_fetch" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_write" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_write.log" 2>&1)
python tools/pmc_to_json.py gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write profiles 2>&1 | tail -3 | tee -a $L
cp profiles/pmc_kernels.json profiles/pmc_match_top2.json gpurun_out/ 2>/dev/null
find gpurun_out/pmc_k_fetch -name "*counter_collection.csv" -exec cp {} gpurun_out/r02_pmc_kernels_FETCH_SIZE.csv \;
find gpurun_out/pmc_k_write -name "*counter_collection.csv" -exec cp {} gpurun_out/r02_pmc_kernels_WRITE_SIZE.csv \;
echo "== bench (default) ==" | tee -a $L
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r02_bench.json
python -c "import json; d=json.load(open('gpurun_out/r02_bench.json')); print('value', d['value'], 'dropin', d['dropin_surface']['value'], 'roofline', d['roofline'], 'cpu', d['cpu_baseline'])" | cut -c1-600 | tee -a $L
echo "== bench --no-cache ==" | tee -a $L
timeout 300 python bench.py --no-cache --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 > gpurun_out/r02_bench_nocache.json
python -c "import json; d=json.load(open('gpurun_out/r02_bench_nocache.json')); print('nocache value', d['value'])" | tee -a $L
echo "== rocprof ==" | tee -a $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline --no-kernels --no-dropin > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200 | tee -a $L
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 > gpurun_out/r02_trace_analysis.txt 2>&1
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r02_trace_by_shape.txt 2>&1
head -24 gpurun_out/r02_trace_analysis.txt | tee -a $L
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r02_bench_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof/bench_domain_stats.csv gpurun_out/r02_bench_domain_stats.csv 2>/dev/null
rm -f gpurun_out/prof/*.db
