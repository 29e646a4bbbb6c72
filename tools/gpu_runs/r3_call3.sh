# round 3, call 3: resblock24 with the reordered prologue + earlier prefetch; per-shape anatomy of the frame with the new kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2))'
echo "== resblock tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "resblock" 2>&1 | tail -3
echo "== microbench"; timeout 300 python tools/bench_resblock.py 2>&1 | grep resblock | grep -v "4 waves"
echo "== probe"; timeout 300 python tools/probe_resblock24.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_probe_resblock24_v2.txt | grep -A12 "== LR (\|2x 2nd"
echo "== IR state + e2e"; timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3
for i in 1 2 3; do
echo "== bench (rb24) $i"; timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels 2>&1 | tail -1 | python -c "$fmt"
echo "== bench (generic lean) $i"; REFVSR_NO_RB24=1 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels 2>&1 | tail -1 | python -c "$fmt"
done
echo "== rocprof trace of the bench"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline --no-kernels --no-dropin > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 > gpurun_out/r3_call3_trace_analysis.txt 2>&1
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r3_call3_trace_by_shape.txt 2>&1
head -30 gpurun_out/r3_call3_trace_analysis.txt
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r3_call3_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof
