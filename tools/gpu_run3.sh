#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest gpu ==" | tee gpurun_out/run3.log
timeout 900 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider -n 2 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "^(FAILED|ERROR|SKIPPED)|passed|failed" gpurun_out/pytest_gpu_full.log | tail -30 | tee -a gpurun_out/run3.log
grep -E "^E  " gpurun_out/pytest_gpu_full.log | head -40 | tee -a gpurun_out/run3.log
grep -E "resblock" gpurun_out/gpu_ops_report.txt | tee -a gpurun_out/run3.log
echo "== kernel micro-benchmarks ==" | tee -a gpurun_out/run3.log
timeout 300 python tools/bench_kernels.py 2>&1 | tail -45 | tee -a gpurun_out/run3.log
echo "== bench ==" | tee -a gpurun_out/run3.log
timeout 600 python bench.py --steps 20 --warmup 3 --cpu-baseline-timeout 200 2>&1 | tail -1 | tee gpurun_out/bench.log
echo "== bench unfused (A/B) ==" | tee -a gpurun_out/run3.log
REFVSR_NO_FUSE=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 | tee gpurun_out/bench_unfused.log
echo "== rocprof ==" | tee -a gpurun_out/run3.log
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -14 "$f" | cut -c1-200; done
echo "== pmc (HBM traffic of match_top2; separate passes) ==" | tee -a gpurun_out/run3.log
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_fetch" -o m -- python "$OLDPWD/tools/pmc_match.py" > "$OLDPWD/gpurun_out/pmc_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_write" -o m -- python "$OLDPWD/tools/pmc_match.py" > "$OLDPWD/gpurun_out/pmc_write.log" 2>&1)
find gpurun_out/pmc_fetch gpurun_out/pmc_write -type f | head
for f in $(find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*counter_collection*.csv"); do echo $f; head -1 $f | cut -c1-300; grep match_top2 $f | head -4 | cut -c1-400; done
