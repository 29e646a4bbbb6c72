#!/bin/bash
# Round-2 opener: is the chained two-ResBlock kernel faster end to end?  (results are bit-identical by construction;
# the fixture / pipelined / determinism tests are re-run with it switched on)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/chain_ab.log
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; match", round(d["roofline"]["mean_launch_ms"],3),"ms", round(d["roofline"]["frac"],3))'
: > $L
for mode in 0 1 2 0 1 2; do
  echo "== bench REFVSR_RESBLOCK_CHAIN=$mode ==" | tee -a $L
  REFVSR_RESBLOCK_CHAIN=$mode timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
# HIP maps the engine's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues: in the round-1 trace the preparation
# stream (P, with the matching kernel) and the forward-branch stream (F) shared one queue and therefore ran in order
for q in 4 8; do
  echo "== bench GPU_MAX_HW_QUEUES=$q ==" | tee -a $L
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
for mode in 1 2; do
  echo "== tests with REFVSR_RESBLOCK_CHAIN=$mode ==" | tee -a $L
  REFVSR_RESBLOCK_CHAIN=$mode timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -n 3 \
      -k "reference_fixture or pipelined or deterministic or two_phase" 2>&1 | tail -3 | tee -a $L
done
