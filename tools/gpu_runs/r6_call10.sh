timeout 500 python bench.py --no-other-configs --no-cpu-baseline --no-wavefront > gpurun_out/r06_bench_call10.json 2> gpurun_out/r06_bench_call10.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_full.json')); r = d['roofline']
print(d['value'], r['traffic'], r['traffic_static'], r.get('live_pmc_error'), r['traffic_source'][:70])
m = d['roofline_match_top2']; print(m['traffic'], m['traffic_static'])
print([(k['kernel'][:24], k.get('traffic_bytes'), k.get('traffic')) for k in d['kernels']][:4])
PY
wc -l gpurun_out/r06_bench_call10.json; tail -2 gpurun_out/r06_bench_call10.err
