#!/bin/bash
# round 5, final tree: the driver's three commands in the driver's form
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r05_final_tree_run2.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== python -m pytest tests/ -x -q -m gpu ==" | tee -a $L
( time timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) 2>&1 | tail -8 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r05_gpu_parity_report2.txt 2>/dev/null
echo "== smoke ==" | tee -a $L
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $L
echo "== python bench.py --gpus 1 --steps 20 --warmup 5 ==" | tee -a $L
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_final_tree2.json 2> gpurun_out/_b.err ) 2>&1 | grep real | tee -a $L
cat gpurun_out/r05_bench_final_tree2.json | tee -a $L
