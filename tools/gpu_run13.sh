#!/bin/bash
# A/B: ab_old/ (previous commit, built) vs the working tree: kernel microbench + bench, then the GPU test suite.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/run14.log
echo "== microbench OLD ==" | tee $L
(cd ab_old && timeout 300 python tools/bench_kernels.py 2>&1 | grep -E "^conv|^resblock" ) > gpurun_out/bk_old.txt
echo "== microbench NEW ==" | tee -a $L
timeout 300 python tools/bench_kernels.py 2>&1 | grep -E "^conv|^resblock" > gpurun_out/bk_new.txt
paste -d'|' gpurun_out/bk_old.txt gpurun_out/bk_new.txt | awk -F'|' '{printf "%-62s | %s\n", substr($1,1,62), substr($2,37,30)}' | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; match", round(d["roofline"]["mean_launch_ms"],3),"ms", round(d["roofline"]["frac"],3))'
echo "== bench OLD ==" | tee -a $L
(cd ab_old && timeout 600 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt") | tee -a $L
echo "== bench NEW ==" | tee -a $L
timeout 600 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "== pytest ==" | tee -a $L
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -n 4 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_full.log | tail -20 | tee -a $L
grep -E "^E  " gpurun_out/pytest_gpu_full.log | head -30 | tee -a $L
