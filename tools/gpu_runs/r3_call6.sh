# round 3, call 6: conv24 (specialised 24-output-channel 3x3 convs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2))'
echo "== conv24 tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "conv24" 2>&1 | tail -5
grep conv24 gpurun_out/gpu_ops_report.txt | tail -16
echo "== full suite"; timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
for i in 1 2; do
echo "== bench (conv24) $i"; timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 | python -c "$fmt"
echo "== bench (generic convs) $i"; REFVSR_NO_CONV24=1 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 | python -c "$fmt"
done
echo "== rocprof trace of the bench"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline --no-kernels --no-dropin --no-wavefront > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 > gpurun_out/r3_call6_trace_analysis.txt 2>&1
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r3_call6_trace_by_shape.txt 2>&1
head -26 gpurun_out/r3_call6_trace_analysis.txt
grep "conv24" gpurun_out/r3_call6_trace_by_shape.txt | head -20
rm -rf gpurun_out/prof
