#!/bin/bash
# round 6, call 11: groups in flight (REFVSR_PIPE_DEPTH - 1) -- the sharded executor issues a whole shard ahead and is as fast as forward_group
mkdir -p gpurun_out
for d in 3 4 6 3 4 6; do
  REFVSR_PIPE_DEPTH=$d timeout 200 python bench.py --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront --no-dropin --no-live-pmc --full-json gpurun_out/_d.json > /dev/null 2> gpurun_out/_d.err
  python -c "
import json; d = json.load(open('gpurun_out/_d.json')); print('REFVSR_PIPE_DEPTH=$d', round(d['value'], 2), d['samples'], d['streams']['median_pass'])" | tee -a gpurun_out/r06_pipe_depth_ab.txt
done
