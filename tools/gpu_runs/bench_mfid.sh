#!/bin/bash
# Config 3 of SURVEY.md 8(d): RefVSR_MFID geometry (C=48, 30 blocks) on one GPU.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 80 python bench.py --config config_RefVSR_MFID --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_MFID.log
cut -c1-330 gpurun_out/bench_MFID.log
