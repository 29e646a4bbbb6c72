#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r2_call2.log
: > $L
echo "== probe ==" | tee -a $L
timeout 300 python tools/probe_resblock.py 2>&1 | tee -a $L
echo "== new tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -n 3 \
    -k "static_input or weight_reload or pipelined or deterministic or api_contract" 2>&1 | tail -8 | tee -a $L
