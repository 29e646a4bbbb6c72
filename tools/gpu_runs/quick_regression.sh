#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -n 3 -k "reference_fixture and not full_size or pipelined or two_phase" > gpurun_out/pytest_quick.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_quick.log | tail; grep -E "^E  " gpurun_out/pytest_quick.log | head -20 | cut -c1-300
