#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu e2e ==" | tee gpurun_out/run7.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -n 2 2>&1 | tail -3 | tee -a gpurun_out/run7.log
for v in 4 3; do for p in 0 1; do
  echo "== bench match v$v, prepare-overlap $((1-p)) ==" | tee -a gpurun_out/run7.log
  if [ $p = 1 ]; then export REFVSR_NO_OVERLAP_PREPARE=1; else unset REFVSR_NO_OVERLAP_PREPARE; fi
  REFVSR_MATCH_VARIANT=$v timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'fps', round(d['ms_per_step'],3),'ms; match', round(d['roofline']['mean_launch_ms'],3),'ms')" | tee -a gpurun_out/run7.log
done; done
