#!/bin/bash
# round 4, call 18 (the last GPU seconds): refvsr_conv1x1_f32 -- the matching's 1x1 map on its own kernel: op test, the matching /
# fixture tests with it on, frame A/B against the generic conv's fp32 mode
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call18.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 45 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 40 --timeout-method=thread -x -k "conv1x1 or feature_match or match_fused or stream_against_reference_fixture or full_size" > gpurun_out/_t.out 2>&1
grep -i -A10 "Traceback\|^E " gpurun_out/_t.out | head -30 | cut -c1-300 | tee -a $L
tail -2 gpurun_out/_t.out | tee -a $L
grep "conv1x1" gpurun_out/gpu_ops_report.txt | tee -a $L
fmt='import sys,json
d=json.loads(sys.stdin.read())
print("value %.1f  samples %s" % (d["value"], d["samples"]))'
B="python bench.py --steps 20 --warmup 5 --repeats 3 --warm-seconds 0.3 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --no-dropin"
echo "== default (map on its own kernel) ==" | tee -a $L
timeout 25 $B 2>/dev/null | tail -1 | python -c "$fmt" | tee -a $L
echo "== REFVSR_NO_MAP1X1=1 ==" | tee -a $L
REFVSR_NO_MAP1X1=1 timeout 25 $B 2>/dev/null | tail -1 | python -c "$fmt" | tee -a $L
