#!/bin/bash
# C = 48 parity depth (64x96 live oracle, full-size reference fixture), RefVSR_IR_MFID kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3_call14.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -k "mfid" 2>&1 | tail -12 | tee -a $L
grep -i "MFID" gpurun_out/gpu_ops_report.txt | tee -a $L
echo "== rocprof trace of the RefVSR_IR_MFID bench ==" | tee -a $L
rm -rf gpurun_out/prof_ir
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_ir" -o bench -- python "$OLDPWD/bench.py" --config config_RefVSR_IR_MFID --steps 10 --warmup 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront > "$OLDPWD/gpurun_out/rocprof_ir.log" 2>&1)
tail -1 gpurun_out/rocprof_ir.log | cut -c1-200 | tee -a $L
python tools/trace_by_shape.py gpurun_out/prof_ir/bench_kernel_trace.csv 60 > gpurun_out/r03_trace_by_shape_IR_MFID.txt 2>&1
head -40 gpurun_out/r03_trace_by_shape_IR_MFID.txt | cut -c1-170 | tee -a $L
rm -rf gpurun_out/prof_ir
