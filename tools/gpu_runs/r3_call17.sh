#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for sz in 270x480 540x960 135x240; do timeout 120 python tools/bench_two_streams.py $sz 2>&1 | grep chains; done | tee gpurun_out/r3_call17.log
PROBE_WAVES=8 timeout 200 python tools/probe_resblock24.py 2>&1 | grep "==" | tee -a gpurun_out/r3_call17.log
