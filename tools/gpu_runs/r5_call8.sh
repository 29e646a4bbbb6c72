#!/bin/bash
# round 5, call 8: full GPU suite after the trim (fused warp, wave priority, sc1 stores, the round-1 wide block kernel, three engine
# knobs gone) + default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call8.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest -m gpu ==" | tee -a $L
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 600 --timeout-method=thread -x 2>&1 | tail -15 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r05_gpu_parity_report.txt 2>/dev/null
echo "== smoke ==" | tee -a $L
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $L
echo "== default bench ==" | tee -a $L
timeout 900 python bench.py --steps 20 --warmup 5 --full-json gpurun_out/r5_bench_full.json > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
echo "rc $? line bytes $(wc -c < gpurun_out/r5_bench.json)" | tee -a $L
cat gpurun_out/r5_bench.json | cut -c1-1800 | tee -a $L
