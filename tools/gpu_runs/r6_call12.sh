#!/bin/bash
# round 6, call 12: match_refine evaluates only the candidates that can win -- matching tests, full-size fixtures, kernel time
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -x -q -k "match or full_size or near_tie or stream_against or midsize" 2>&1 | tail -6
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs --no-live-pmc > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
st=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
grep -E "match_" $st | cut -c1-140
rm -rf gpurun_out/prof
