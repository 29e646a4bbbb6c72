#!/bin/bash
# one-workgroup-per-CU resident convs: 8 waves x 2 groups (8x32 tile) vs 8 x 4 and 16 x 2 (16x32 tile)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_onewg.log
: > $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
for m in 8x2 8x4 16x2 8x2 8x4 16x2; do
  export REFVSR_CONV_ONEWG=$m
  echo "== MFID $m ==" | tee -a $L
  timeout 300 python bench.py --config config_RefVSR_MFID --steps 12 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
for m in 8x2 8x4 16x2; do
  export REFVSR_CONV_ONEWG=$m
  echo "== MFID_8K 1080p $m ==" | tee -a $L
  timeout 600 python bench.py --config config_RefVSR_MFID_8K --size 1080x1920 --frames 5 --steps 4 --warmup 1 --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
for m in 8x4 16x2; do
  export REFVSR_CONV_ONEWG=$m
  echo "== tests $m ==" | tee -a $L
  timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x -k "HD48 or 8k or hd" 2>&1 | tail -3 | tee -a $L
done
