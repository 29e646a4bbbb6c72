#!/bin/bash
# GPU session 2: host probe -> tests -> kernel micro-benchmarks -> bench (with cpu baseline) -> rocprofv3 (csv).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/gpu_ops_report.txt
echo "== host probe ==" | tee gpurun_out/run2.log
python - <<'PY' 2>&1 | tee -a gpurun_out/run2.log
import os
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
PY
echo "== pytest gpu ==" | tee -a gpurun_out/run2.log
timeout 900 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider -n 2 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "^(PASSED|FAILED|ERROR|SKIPPED)|passed|failed" gpurun_out/pytest_gpu_full.log | tail -70 | tee -a gpurun_out/run2.log
grep -E "^E  " gpurun_out/pytest_gpu_full.log | head -40 | tee -a gpurun_out/run2.log
echo "== kernel micro-benchmarks ==" | tee -a gpurun_out/run2.log
timeout 300 python tools/bench_kernels.py 2>&1 | tail -40 | tee -a gpurun_out/run2.log
echo "== bench ==" | tee -a gpurun_out/run2.log
timeout 600 python bench.py --steps 20 --warmup 3 --cpu-baseline-timeout 200 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== bench no-cache (as the reference executes) ==" | tee -a gpurun_out/run2.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cache --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_nocache.log
echo "== rocprof ==" | tee -a gpurun_out/run2.log
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-400
find gpurun_out/prof -type f | head
for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -30 "$f"; done
