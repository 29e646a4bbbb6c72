#!/bin/bash
# exact matching + new parity tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_call6.log
: > $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; match", round(d["roofline"]["mean_launch_ms"],3),"ms", round(d["roofline"]["frac"],3))'
rm -f gpurun_out/gpu_ops_report.txt
echo "== match tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -k "match" 2>&1 | tail -15 | tee -a $L
echo "== parity tests ==" | tee -a $L
timeout 1200 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -n 4 -k "full_size_against or long_recurrence or reference_fixture or midsize" 2>&1 | tail -15 | tee -a $L
grep -E "match|full-size|recurrence" gpurun_out/gpu_ops_report.txt | tee -a $L
echo "== bench ==" | tee -a $L
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
timeout 300 python tools/bench_kernels.py 2>&1 | grep -E "^match" | tee -a $L
