#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest gpu ==" | tee gpurun_out/run6.log
timeout 900 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider -n 2 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "^(FAILED|ERROR|SKIPPED)|passed|failed" gpurun_out/pytest_gpu_full.log | tail -30 | tee -a gpurun_out/run6.log
grep -E "^E  " gpurun_out/pytest_gpu_full.log | head -40 | tee -a gpurun_out/run6.log
echo "== bench (overlap on) ==" | tee -a gpurun_out/run6.log
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee gpurun_out/bench.log
echo "== bench (overlap off) ==" | tee -a gpurun_out/run6.log
REFVSR_NO_OVERLAP=1 timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee gpurun_out/bench_nooverlap.log
echo "== bench match v3 (78 KB LDS) + overlap ==" | tee -a gpurun_out/run6.log
REFVSR_MATCH_VARIANT=3 timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee gpurun_out/bench_v3.log
