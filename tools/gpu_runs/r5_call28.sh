#!/bin/bash
# round 5, call 28: roll-over windows INSIDE the frame groups (only their forward branch is the long one): bit-identity + same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r05_pipelined_restart_ab.txt
echo "== roll-over windows inside the groups (default) vs REFVSR_SERIAL_RESTART=1 (round 4: drain + reference order on M, the window alone) ==" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -p no:cacheprovider -k "frame_groups or groups_of_any or batch_samples or cli_frame" 2>&1 | tail -3 | tee -a $L
timeout 600 python tools/soak.py --frames 300 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $L
for rep in 1 2; do
for steps in 20 100; do
for serial in 1 ""; do
  REFVSR_SERIAL_RESTART=$serial timeout 600 python bench.py --steps $steps --warmup 5 --repeats 3 --no-dropin --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront \
      --full-json gpurun_out/_rs_full.json > gpurun_out/_rs.json 2> gpurun_out/_rs.err
  python - <<PY | tee -a $L
import json
j=json.load(open('gpurun_out/_rs.json'))
print('rep $rep steps $steps serial_restart=[$serial]: groups', round(j['value'],2), j['samples'])
PY
done
done
done
