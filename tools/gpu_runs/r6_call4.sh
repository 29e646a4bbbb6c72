#!/bin/bash
# round 6, call 4: phase-A groups split over P | M in the sharded executor
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "phase_a_group or two_process or bench_gpus_2 or context_exchange_block" 2>&1 | tail -15
cp /tmp/bench_n2_configs3_full.json gpurun_out/r06_bench_n2_configs3_full.json 2>/dev/null
grep "configs\[3\]" gpurun_out/gpu_ops_report.txt | tail -2
timeout 400 python bench.py --no-other-configs --no-cpu-baseline --no-kernels --no-dropin > gpurun_out/r06_bench_call4.json 2> gpurun_out/r06_bench_call4.err; tail -c 1800 gpurun_out/r06_bench_call4.json; tail -5 gpurun_out/r06_bench_call4.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_full.json'))
wm = d.get('wavefront_model', {})
for k in ('one_rank_same_clip', 'one_rank_wavefront', 'one_rank', 'phase_ms_per_frame_measured'):
    print(k, wm.get(k))
PY
