#!/bin/bash
# round 5, call 1: multi-map launches (ABI 11) -- bit-identity of every batched entry point against its single-map call, regression
# of the kernels that gained the batch path (resblock24, conv24 family, warps), and the per-map cost with B = 1..4 maps per launch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call1.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== op tests (multimap + the kernels it touched) ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider --timeout 200 --timeout-method=thread -x \
  -k "multimap or resblock24 or conv24 or conf_alpha or warp or shuffle2 or conv_last or conv48 or conv32 or compute_up or conv_stacks" 2>&1 | tail -12 | tee -a $L
echo "== multimap microbench ==" | tee -a $L
timeout 300 python tools/bench_multimap.py 2>&1 | grep "^multimap" | tee -a $L
cp $L gpurun_out/r5_call1_done.log
