#!/bin/bash
# device-side durations of the microbenchmark kernels (the Python timing loop is host-bound below ~12 us per launch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_micro
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_micro" -o micro -- python "$OLDPWD/tools/bench_kernels.py" > "$OLDPWD/gpurun_out/microbench_rocprof.log" 2>&1)
grep -E "^conv|^resblock|^match" gpurun_out/microbench_rocprof.log | cut -c1-120
head -30 gpurun_out/prof_micro/micro_kernel_stats.csv | cut -c1-150
