#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest gpu ==" | tee gpurun_out/run4.log
timeout 900 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider -n 2 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "^(FAILED|ERROR|SKIPPED)|passed|failed" gpurun_out/pytest_gpu_full.log | tail -30 | tee -a gpurun_out/run4.log
grep -E "^E  " gpurun_out/pytest_gpu_full.log | head -40 | tee -a gpurun_out/run4.log
grep -E "HD|gather|conv_direct" gpurun_out/gpu_ops_report.txt | tee -a gpurun_out/run4.log
echo "== kernel micro-benchmarks ==" | tee -a gpurun_out/run4.log
timeout 300 python tools/bench_kernels.py 2>&1 | grep -E "match|resblock|floor|vgg" | tee -a gpurun_out/run4.log
echo "== bench ==" | tee -a gpurun_out/run4.log
timeout 600 python bench.py --steps 20 --warmup 3 --cpu-baseline-timeout 200 2>&1 | tail -1 | tee gpurun_out/bench.log
echo "== bench RefVSR_MFID (configs[2] geometry) ==" | tee -a gpurun_out/run4.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --config config_RefVSR_MFID 2>&1 | tail -1 | cut -c1-420 | tee gpurun_out/bench_MFID.log
echo "== rocprof ==" | tee -a gpurun_out/run4.log
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -16 "$f" | cut -c1-160; done
