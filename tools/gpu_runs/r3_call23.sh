#!/bin/bash
# bench without the ResBlock-run event records in the timed pass: copyBuffer launches per frame in the kernel trace, frames/s
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3_call23.log
: > $L
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-kernels --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'fps', round(d['ms_per_step'],3),'ms', 'resblock', round(d['roofline']['mean_launch_ms']*1e3,2), 'us', round(d['roofline']['frac'],4))" | tee -a $L; done
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline --no-kernels --no-dropin --no-wavefront > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 2>&1 | grep -i "copyBuffer\|fillBuffer\|at::native" | cut -c1-160 | tee -a $L
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 2>&1 | head -4 | tee -a $L
rm -rf gpurun_out/prof
