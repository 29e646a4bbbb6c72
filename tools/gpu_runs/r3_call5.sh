# round 3, call 5: batched staging of the streamed convs, fused warp
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2))'
echo "== conv / warp tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "conv or spynet or warp" 2>&1 | tail -5
echo "== full suite"; timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
for i in 1 2; do
echo "== bench (fused warp) $i"; REFVSR_FUSE_WARP=1 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 | python -c "$fmt"
echo "== bench (separate warp) $i"; timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 | python -c "$fmt"
done
echo "== rocprof trace of the bench"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline --no-kernels --no-dropin --no-wavefront > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 > gpurun_out/r3_call5_trace_analysis.txt 2>&1
python tools/trace_by_shape.py gpurun_out/prof/bench_kernel_trace.csv 300 > gpurun_out/r3_call5_trace_by_shape.txt 2>&1
head -24 gpurun_out/r3_call5_trace_analysis.txt
grep "false, false, false\|warp\|copyBuffer" gpurun_out/r3_call5_trace_by_shape.txt | head -40
# what precedes the one-workgroup copyBuffer launches?
python - <<'P'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof/bench_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
import collections
prev = collections.Counter()
for i, r in enumerate(rows):
    if 'copyBuffer' in r['Kernel_Name'] and int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', 0)) <= 512:
        q = r.get('Queue_Id')
        j = i - 1
        while j >= 0 and rows[j].get('Queue_Id') != q:
            j -= 1
        k = i + 1
        while k < len(rows) and rows[k].get('Queue_Id') != q:
            k += 1
        prev[(rows[j]['Kernel_Name'][:60] if j >= 0 else '-', rows[k]['Kernel_Name'][:60] if k < len(rows) else '-')] += 1
for k, v in prev.most_common(12):
    print(v, k)
P
rm -rf gpurun_out/prof
