mkdir -p gpurun_out
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs --no-live-pmc > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
st=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
python - "$st" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'match_' in r['Name']:
        print(r['Name'][:40], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us', r['Percentage'], '%')
PY
rm -rf gpurun_out/prof
