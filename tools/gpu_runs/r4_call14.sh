#!/bin/bash
# round 4, call 14: kernel trace of a RefVSR_IR_MFID frame (where do its 38 ms go?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call14.log
: > $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --config config_RefVSR_IR_MFID --steps 8 --warmup 2 --repeats 1 --no-pipeline --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r04_trace_by_shape_IR.txt 2>&1
head -45 gpurun_out/r04_trace_by_shape_IR.txt | cut -c1-175 | tee -a $L
rm -rf gpurun_out/prof
