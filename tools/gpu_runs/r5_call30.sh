#!/bin/bash
# round 5, call 30: groups in flight (REFVSR_PIPE_DEPTH 3 = two groups, 4 = three) over 100-step passes, after the roll-over change
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r05_pipe_depth_ab.txt
: > $L
for rep in 1 2; do
for d in 3 4; do
  REFVSR_PIPE_DEPTH=$d timeout 600 python bench.py --steps 100 --warmup 5 --repeats 3 --no-dropin --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront \
      --full-json gpurun_out/_pd_full.json > gpurun_out/_pd.json 2> gpurun_out/_pd.err
  python - <<PY | tee -a $L
import json
j=json.load(open('gpurun_out/_pd.json'))
print('rep $rep REFVSR_PIPE_DEPTH=$d: groups', round(j['value'],2), j['samples'])
PY
done
done
