#!/bin/bash
# round 5, call 23: what the Python collector costs a plain loop (bench with and without gc.freeze, 100 timed steps per pass); p_conv.0 default (mt by stride)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r05_gc_freeze_ab.txt
: > $L
for rep in 1 2; do
for fl in "" "--no-gc-freeze"; do
  timeout 600 python bench.py --steps 100 --warmup 5 --repeats 3 --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront $fl \
      --full-json gpurun_out/_gc_full.json > gpurun_out/_gc.json 2> gpurun_out/_gc.err
  python - <<PY | tee -a $L
import json
j=json.load(open('gpurun_out/_gc.json'))
print('flags [$fl] rep $rep: groups', round(j['value'],2), j['samples'], ' per-call', j['one_frame_per_call']['value'], j['one_frame_per_call']['samples'], ' dropin', j['dropin_surface']['value'], j['dropin_surface']['samples'])
PY
done
done
