#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest gpu ==" | tee gpurun_out/run8.log
timeout 1200 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider -n 2 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "^(FAILED|ERROR|SKIPPED)|passed|failed" gpurun_out/pytest_gpu_full.log | tail -30 | tee -a gpurun_out/run8.log
grep -E "^E  " gpurun_out/pytest_gpu_full.log | head -40 | tee -a gpurun_out/run8.log
echo "== kernel micro-benchmarks ==" | tee -a gpurun_out/run8.log
timeout 300 python tools/bench_kernels.py 2>&1 | grep -E "resblock|spynet|5x5|floor|2x 24" | tee -a gpurun_out/run8.log
echo "== bench ==" | tee -a gpurun_out/run8.log
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'fps', round(d['ms_per_step'],3),'ms; match', round(d['roofline']['mean_launch_ms'],3),'ms', round(d['roofline']['frac'],3))" | tee -a gpurun_out/run8.log
echo "== rocprof ==" | tee -a gpurun_out/run8.log
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -12 "$f" | cut -c1-150; done
