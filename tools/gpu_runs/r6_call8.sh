#!/bin/bash
# round 6, call 8: the HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no tracing) on this round's tree -> pmc_kernels.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_fetch" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_write" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_write.log" 2>&1)
python tools/pmc_to_json.py gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write gpurun_out 2>&1 | tail -3
find gpurun_out/pmc_k_fetch -name "*counter_collection.csv" -exec cp {} gpurun_out/r06_pmc_kernels_FETCH_SIZE.csv \;
find gpurun_out/pmc_k_write -name "*counter_collection.csv" -exec cp {} gpurun_out/r06_pmc_kernels_WRITE_SIZE.csv \;
rm -rf gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write
python -c "
import json; d=json.load(open('gpurun_out/pmc_kernels.json')); print({k: round(v/1e6,2) for k,v in d['traffic_bytes_per_launch'].items()})"
tail -3 gpurun_out/pmc_k_fetch.log
