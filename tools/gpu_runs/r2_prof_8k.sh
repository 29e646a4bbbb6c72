#!/bin/bash
# kernel-time anatomy of BASELINE configs[4] (RefVSR_MFID_8K, 1080x1920 -> 4320x7680, t = 5) and configs[2] (RefVSR_MFID 270p) on one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfgsz in "config_RefVSR_MFID_8K 1080x1920 4 1" "config_RefVSR_MFID 270x480 10 2"; do
  set -- $cfgsz
  rm -rf gpurun_out/prof_$1
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_$1" -o b -- python "$OLDPWD/bench.py" --config $1 --size $2 --frames 5 --steps $3 --warmup $4 --no-cpu-baseline --no-kernels --no-dropin > "$OLDPWD/gpurun_out/prof_$1.log" 2>&1)
  tail -1 gpurun_out/prof_$1.log | cut -c1-300
  python tools/trace_by_shape.py gpurun_out/prof_$1/b_kernel_trace.csv 40 > gpurun_out/r02_trace_by_shape_$1.txt 2>&1
  head -32 gpurun_out/r02_trace_by_shape_$1.txt | cut -c1-190
  rm -f gpurun_out/prof_$1/*.db gpurun_out/prof_$1/*trace.csv
done
