#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== host profile ==" | tee gpurun_out/run12.log
python tools/host_profile.py 2>&1 | grep "host enqueue" | tee -a gpurun_out/run12.log
echo "== pytest (ops + e2e) ==" | tee -a gpurun_out/run12.log
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -n 2 2>&1 | tail -4 | tee -a gpurun_out/run12.log
for mode in "" "--no-pipeline" "--no-frame-ids"; do
  echo "== bench $mode ==" | tee -a gpurun_out/run12.log
  timeout 600 python bench.py --steps 40 --warmup 3 --no-cpu-baseline $mode 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'fps', round(d['ms_per_step'],3),'ms; match', round(d['roofline']['mean_launch_ms'],3),'ms', round(d['roofline']['frac'],3))" | tee -a gpurun_out/run12.log
done
