#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/conv_ab.log
echo "== pytest ops + fixtures ==" | tee $L
timeout 600 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -n 4 -k "not live_oracle and not two_process and not full_size" > gpurun_out/pytest_gpu_part.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu_part.log | tail -20 | tee -a $L
grep -E "^E  " gpurun_out/pytest_gpu_part.log | head -30 | cut -c1-300 | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; match", round(d["roofline"]["mean_launch_ms"],3),"ms", round(d["roofline"]["frac"],3))'
for envs in "A=1" "REFVSR_CONV_TILES=2" "REFVSR_CONV_NO_PREFETCH=1" "REFVSR_CONV_RES_MAX=8" "A=2"; do
  echo "== bench [$envs] ==" | tee -a $L
  env $envs timeout 600 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
for envs in "A=1" "REFVSR_CONV_TILES=2"; do
  echo "== microbench [$envs] ==" | tee -a $L
  env $envs timeout 300 python tools/bench_kernels.py 2>&1 | grep -E "^conv (2x|HR|LR 24\+)|^resblock 2x" | tee -a $L
done
