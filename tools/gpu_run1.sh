#!/bin/bash
# First GPU session: smoke -> GPU test suite -> bench -> rocprofv3 kernel stats.  Everything under timeouts.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/gpu_ops_report.txt
echo "== device ==" | tee gpurun_out/run1.log
timeout 300 python -c "import torch; print(torch.cuda.get_device_name(0), torch.version.hip); import os; print('cpus', os.cpu_count())" 2>&1 | tee -a gpurun_out/run1.log
echo "== smoke ==" | tee -a gpurun_out/run1.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -30 | tee -a gpurun_out/run1.log
echo "== pytest gpu ==" | tee -a gpurun_out/run1.log
timeout 1500 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider -n 2 2>&1 | tail -250 > gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
echo "== bench ==" | tee -a gpurun_out/run1.log
timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | tail -5 | tee gpurun_out/bench.log
echo "== rocprof ==" | tee -a gpurun_out/run1.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
ls -la gpurun_out/prof 2>/dev/null | head; find gpurun_out/prof -name "*stats*" | head
for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -40 "$f"; done
