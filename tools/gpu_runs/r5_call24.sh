#!/bin/bash
# round 5, call 24: groups cut at the restarts of the forward branch (Network.steady_windows_ahead): test + bench at 20 and 100 steps per pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r05_group_cut_ab.txt
: > $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -p no:cacheprovider -k "steady_windows_ahead or cli_frame_groups or frame_groups_are" 2>&1 | tail -3 | tee -a $L
for steps in 20 100; do
  timeout 600 python bench.py --steps $steps --warmup 5 --repeats 5 --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront \
      --full-json gpurun_out/_gcut_full.json > gpurun_out/_gcut.json 2> gpurun_out/_gcut.err
  tail -2 gpurun_out/_gcut.err | grep -v amdgpu | tee -a $L
  python - <<PY | tee -a $L
import json
j=json.load(open('gpurun_out/_gcut.json'))
print('steps $steps: groups', round(j['value'],2), j['samples'], ' per-call', j['one_frame_per_call']['value'], ' dropin', j['dropin_surface']['value'], ' pcie', j['pcie_inclusive']['value'], ' frac', (j.get('roofline') or {}).get('frac'))
PY
done
