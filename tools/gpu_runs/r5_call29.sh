#!/bin/bash
# round 5, call 29: rocprofv3 kernel stats + trace analysis of the bench command on the last tree (group mode)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
f=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
st=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
cp $st gpurun_out/r05_bench_kernel_stats_last_tree.csv
python tools/trace_analysis.py $f 8 20 > gpurun_out/r05_trace_analysis_last_tree.txt 2>&1
python tools/trace_by_shape.py $f 300 > gpurun_out/r05_trace_by_shape_last_tree.txt 2>&1
head -14 gpurun_out/r05_trace_analysis_last_tree.txt
head -12 gpurun_out/r05_bench_kernel_stats_last_tree.csv
rm -rf gpurun_out/prof
