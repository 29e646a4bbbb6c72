#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest conv ops ==" | tee gpurun_out/run9.log
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -n 2 -k "conv or resblock or spynet or stacks" 2>&1 | tail -2 | tee -a gpurun_out/run9.log
echo "== kernel micro-benchmarks ==" | tee -a gpurun_out/run9.log
timeout 300 python tools/bench_kernels.py 2>&1 | grep -E "resblock|spynet|5x5|2x 24|LR 24" | tee -a gpurun_out/run9.log
echo "== bench ==" | tee -a gpurun_out/run9.log
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'fps', round(d['ms_per_step'],3),'ms; match', round(d['roofline']['mean_launch_ms'],3),'ms', round(d['roofline']['frac'],3))" | tee -a gpurun_out/run9.log
