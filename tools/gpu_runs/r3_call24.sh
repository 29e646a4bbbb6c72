#!/bin/bash
# which HIP API calls launch the __amd_rocclr_copyBuffer kernels of a frame?  (kernel trace + HIP API trace, correlation ids)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3_call24.log
: > $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 12 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
ls gpurun_out/prof | tee -a $L
python - <<'PY' | tee -a $L
import csv, collections, glob
kt = glob.glob('gpurun_out/prof/*kernel_trace.csv')[0]
ha = glob.glob('gpurun_out/prof/*hip_api_trace.csv')
print('files', kt, ha)
rows = list(csv.DictReader(open(kt)))
cb = [r for r in rows if 'copyBuffer' in r['Kernel_Name']]
print('copyBuffer dispatches', len(cb), 'of', len(rows))
if ha:
    api = list(csv.DictReader(open(ha[0])))
    print('api columns', list(api[0].keys()))
    by = {r['Correlation_Id']: r for r in api}
    c = collections.Counter()
    for r in cb:
        a = by.get(r['Correlation_Id'])
        c[(a['Function'] if a else 'no api row', r['Grid_Size_X'])] += 1
    for k, v in c.most_common(12):
        print(v, k)
    # api call counts overall
    c2 = collections.Counter(r['Function'] for r in api)
    print(c2.most_common(14))
PY
rm -rf gpurun_out/prof
