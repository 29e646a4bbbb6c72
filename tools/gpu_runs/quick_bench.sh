#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 60 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_quick.log
python -c "import json; d=json.load(open('gpurun_out/bench_quick.log')); print(round(d['value'],2), 'fps; first_frame_ms', d['first_frame_ms'], '; match frac', round(d['roofline']['frac'],3))"
