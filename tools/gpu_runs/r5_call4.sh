#!/bin/bash
# round 5, call 4: why do frame groups (G = 4) not move the frame rate?  rocprofv3 kernel traces of the group mode and of the
# one-frame-per-call mode, per-kernel time per frame
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call4.log
: > $L
for g in 4 1; do
  rm -rf gpurun_out/prof_g$g
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_g$g" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --group $g --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof_g$g.log" 2>&1)
  f=$(find gpurun_out/prof_g$g -name "*kernel_trace.csv" | head -1)
  echo "== group $g: $f ==" | tee -a $L
  tail -1 gpurun_out/rocprof_g$g.log | cut -c1-400 | tee -a $L
  python tools/trace_analysis.py $f 8 20 > gpurun_out/r05_trace_analysis_g$g.txt 2>&1
  python tools/trace_by_shape.py $f 300 > gpurun_out/r05_trace_by_shape_g$g.txt 2>&1
  head -24 gpurun_out/r05_trace_analysis_g$g.txt | tee -a $L
  rm -rf gpurun_out/prof_g$g
done
