#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 170 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -n 3 -k "two_phase or False-True" > gpurun_out/pytest_wavefront.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_wavefront.log | tail; grep -E "^E  " gpurun_out/pytest_wavefront.log | head -20 | cut -c1-300
