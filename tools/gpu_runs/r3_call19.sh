#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3_call19.log
: > $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
B="python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin"
for i in 1 2; do
echo "default" | tee -a $L;                 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "three streams (one M; F, P on their own queues)" | tee -a $L;  REFVSR_PIPE_ONE_M=2 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "8 queues, one M" | tee -a $L;     GPU_MAX_HW_QUEUES=8 REFVSR_PIPE_ONE_M=1 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "P high priority" | tee -a $L;     REFVSR_STREAM_PRIORITY=1 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "3 queues" | tee -a $L;     GPU_MAX_HW_QUEUES=3 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
rm -rf gpurun_out/prof
(cd /tmp && REFVSR_PIPE_ONE_M=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline --no-kernels --no-dropin --no-wavefront > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 2>&1 | grep -A12 "per HIP queue" | cut -c1-330 | tee -a $L
rm -rf gpurun_out/prof
