# round 3, call 7: conv48 (specialised 48-output-channel convs of the MFID family)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2))'
echo "== conv24/48 + torch ops + warp tests"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_torch_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "conv48 or conv24 or torch or fused_warp" 2>&1 | tail -5
grep conv48 gpurun_out/gpu_ops_report.txt | tail -10
echo "== full suite"; timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -6
for i in 1 2; do
echo "== MFID (conv48) $i"; timeout 400 python bench.py --config config_RefVSR_MFID --steps 20 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 | python -c "$fmt"
echo "== MFID (generic) $i"; REFVSR_NO_CONV24=1 timeout 400 python bench.py --config config_RefVSR_MFID --steps 20 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 | python -c "$fmt"
done
echo "== MFID_8K 1080p"; timeout 600 python bench.py --config config_RefVSR_MFID_8K --size 1080x1920 --frames 5 --steps 6 --warmup 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8K', round(d['value'],2), 'fps', round(d['ms_per_step'],1), 'ms')"
echo "== headline"; timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 | python -c "$fmt"
echo "== rocprof trace of the MFID bench"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --config config_RefVSR_MFID --steps 12 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r03_trace_by_shape_MFID.txt 2>&1
head -14 gpurun_out/r03_trace_by_shape_MFID.txt | cut -c1-150
rm -rf gpurun_out/prof
