#!/bin/bash
# 8-wave conv workgroups where LDS admits one workgroup per CU: op tests + same-box A/B (REFVSR_CONV_NO_NW8=1 = before)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_nw8.log
: > $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3 | tee -a $L
for v in 0 1 0 1; do
  echo "== S  NO_NW8=$v ==" | tee -a $L
  if [ $v = 1 ]; then export REFVSR_CONV_NO_NW8=1; else unset REFVSR_CONV_NO_NW8; fi
  timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
for v in 0 1; do
  echo "== MFID NO_NW8=$v ==" | tee -a $L
  if [ $v = 1 ]; then export REFVSR_CONV_NO_NW8=1; else unset REFVSR_CONV_NO_NW8; fi
  timeout 300 python bench.py --config config_RefVSR_MFID --steps 12 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
for v in 0 1; do
  echo "== MFID_8K 1080p NO_NW8=$v ==" | tee -a $L
  if [ $v = 1 ]; then export REFVSR_CONV_NO_NW8=1; else unset REFVSR_CONV_NO_NW8; fi
  timeout 600 python bench.py --config config_RefVSR_MFID_8K --size 1080x1920 --frames 5 --steps 4 --warmup 1 --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
unset REFVSR_CONV_NO_NW8
echo "== stream tests ==" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3 | tee -a $L
