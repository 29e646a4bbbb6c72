#!/bin/bash
# round 5, call 25: same-box A/B of the group cut (fixed groups of four vs groups cut at the restarts), 20 and 100 steps per pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r05_group_cut_ab.txt
echo "== same box: fixed groups (REFVSR_BENCH_FIXED_GROUPS=1) vs groups cut at the restarts ==" | tee -a $L
for rep in 1 2; do
for steps in 20 100; do
for fixed in 1 ""; do
  REFVSR_BENCH_FIXED_GROUPS=$fixed timeout 600 python bench.py --steps $steps --warmup 5 --repeats 3 --no-dropin --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront \
      --full-json gpurun_out/_gcut_full.json > gpurun_out/_gcut.json 2> gpurun_out/_gcut.err
  python - <<PY | tee -a $L
import json
j=json.load(open('gpurun_out/_gcut.json'))
print('rep $rep steps $steps fixed=[$fixed]: groups', round(j['value'],2), j['samples'])
PY
done
done
done
