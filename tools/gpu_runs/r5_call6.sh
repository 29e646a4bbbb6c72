#!/bin/bash
# round 5, call 6: group mode on the P | F | M layout without a backward head (the new defaults): bit-identity tests, the full-size HD
# fixture test (configs[4] at 1080 x 1920 against the reference), default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call6.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== tests ==" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 --timeout-method=thread \
  -k "frame_groups or pipelined_mode or full_size_hd or bench_gpus_2 or context_export or two_process" 2>&1 | tail -15 | tee -a $L
grep "full-size HD" gpurun_out/gpu_ops_report.txt gpurun_out/*.txt 2>/dev/null | tee -a $L
echo "== default bench ==" | tee -a $L
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --full-json gpurun_out/r5_bench_full.json > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
echo "rc $? line bytes $(wc -c < gpurun_out/r5_bench.json)" | tee -a $L
cat gpurun_out/r5_bench.json | cut -c1-3000 | tee -a $L
tail -3 gpurun_out/r5_bench.err | tee -a $L
