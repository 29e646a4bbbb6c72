#!/bin/bash
# Round-2 call 1: chained ResBlock A/B, hardware-queue count A/B, per-shape device durations of the microbenchmarks.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_call1.log
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; match", round(d["roofline"]["mean_launch_ms"],3),"ms", round(d["roofline"]["frac"],3))'
: > $L
for mode in 0 1 2 0 1; do
  echo "== bench REFVSR_RESBLOCK_CHAIN=$mode ==" | tee -a $L
  REFVSR_RESBLOCK_CHAIN=$mode timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
for q in 8 2; do
  echo "== bench GPU_MAX_HW_QUEUES=$q ==" | tee -a $L
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
echo "== bench chain=1 queues=8 ==" | tee -a $L
GPU_MAX_HW_QUEUES=8 REFVSR_RESBLOCK_CHAIN=1 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "== bench --no-pipeline / --no-frame-ids ==" | tee -a $L
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-pipeline 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-frame-ids 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "== microbench under rocprof ==" | tee -a $L
rm -rf gpurun_out/prof_micro
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_micro" -o micro -- python "$OLDPWD/tools/bench_kernels.py" > "$OLDPWD/gpurun_out/microbench_rocprof.log" 2>&1)
grep -E "^conv|^resblock|^match" gpurun_out/microbench_rocprof.log | cut -c1-120 | tee -a $L
python tools/trace_by_shape.py gpurun_out/prof_micro/micro_kernel_trace.csv 200 > gpurun_out/r2_micro_by_shape.txt 2>&1
head -60 gpurun_out/r2_micro_by_shape.txt | cut -c1-170 | tee -a $L
echo "== tests with chain=1 ==" | tee -a $L
REFVSR_RESBLOCK_CHAIN=1 timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -n 3 \
    -k "reference_fixture or pipelined or deterministic or two_phase" 2>&1 | tail -3 | tee -a $L
rm -f gpurun_out/prof_micro/*.db
