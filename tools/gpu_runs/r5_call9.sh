#!/bin/bash
# round 5, call 9: the GPU suite after the trim (second attempt) + PMC passes with the multi-map groups
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call9.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest -m gpu ==" | tee -a $L
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 600 --timeout-method=thread 2>&1 | tail -15 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r05_gpu_parity_report.txt 2>/dev/null
echo "== pmc ==" | tee -a $L
rm -rf gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_fetch" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_write" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_write.log" 2>&1)
python tools/pmc_to_json.py gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write gpurun_out 2>&1 | tail -3 | tee -a $L
find gpurun_out/pmc_k_fetch -name "*counter_collection.csv" -exec cp {} gpurun_out/r05_pmc_kernels_FETCH_SIZE.csv \;
find gpurun_out/pmc_k_write -name "*counter_collection.csv" -exec cp {} gpurun_out/r05_pmc_kernels_WRITE_SIZE.csv \;
rm -rf gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write
python -c "
import json; d=json.load(open('gpurun_out/pmc_kernels.json')); print({k: round(v/1e6,2) for k,v in d['traffic_bytes_per_launch'].items()})" | tee -a $L
