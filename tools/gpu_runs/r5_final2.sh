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
echo "== the same box, 100 timed steps per pass (sustained load) ==" | tee -a $L
timeout 600 python bench.py --steps 100 --warmup 5 --repeats 3 --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront \
    --full-json gpurun_out/_s100_full.json > gpurun_out/r05_bench_final_tree_100_steps.json 2> gpurun_out/_s100.err
python - <<'PY' | tee -a $L
import json
j=json.load(open('gpurun_out/r05_bench_final_tree_100_steps.json'))
print('100 steps: groups', round(j['value'],2), j['samples'], 'per-call', j['one_frame_per_call']['value'], 'dropin', j['dropin_surface']['value'], 'pcie', j['pcie_inclusive']['value'], 'frac', j['roofline']['frac'] if j.get('roofline') else None)
PY
