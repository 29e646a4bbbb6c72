#!/bin/bash
# match_top2 v6 (fragment reads one tile ahead) vs v4: parity tests, micro-benchmark, bench A/B; RefVSR_IR vis test
# (historical: the v6 kernel and its REFVSR_MATCH_TOP2 knob were removed after this run -- slower, profiles/r03_match_top2_v6_ab.txt)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call9.log
: > $L
timeout 600 python -m pytest tests/test_gpu_ops.py -q --no-header -p no:cacheprovider -k "match" 2>&1 | tail -4 | tee -a $L
timeout 300 python -m pytest tests/test_gpu_e2e.py -q --no-header -p no:cacheprovider -k "ir_" 2>&1 | tail -4 | tee -a $L
for i in 1 2; do
timeout 120 python tools/bench_match.py 2>&1 | grep match_top2 | tee -a $L
REFVSR_MATCH_TOP2=4 timeout 120 python tools/bench_match.py 2>&1 | grep match_top2 | tee -a $L
done
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; match", d["roofline_match_top2"]["frac"], d["roofline_match_top2"].get("mean_launch_ms"))'
for i in 1 2; do
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
REFVSR_MATCH_TOP2=4 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
