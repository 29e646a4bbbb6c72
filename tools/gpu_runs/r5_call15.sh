#!/bin/bash
# round 5, call 15: border tiles masked at the park instead of behind their loads (resblock24, conv24 family): op tests, stream tests,
# micro-benchmarks, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call15.log
: > $L
echo "== op tests ==" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider --timeout 300 --timeout-method=thread -x 2>&1 | tail -5 | tee -a $L
echo "== engine tests ==" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 400 --timeout-method=thread -x -k "stream_against_reference or frame_groups or full_size_against or long_stream or round4" 2>&1 | tail -4 | tee -a $L
echo "== multimap microbench ==" | tee -a $L
timeout 300 python tools/bench_multimap.py 2>&1 | grep "^multimap" | tee -a $L
fmt='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f  samples %s  percall %s dropin %s roofline %s" % (d["value"], d["samples"], d.get("one_frame_per_call") and d["one_frame_per_call"]["value"], d.get("dropin_surface") and d["dropin_surface"]["value"], d["roofline"] and (round(d["roofline"].get("frac"),4), d["roofline"].get("mean_launch_ms"))))'
echo "== bench ==" | tee -a $L
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --full-json gpurun_out/_b.json | python -c "$fmt" | tee -a $L; done
