#!/bin/bash
# round 5, call 14: the 12-frame full-size stream against the reference fixture (north-star bar on every frame, per-frame calls and
# frame groups) + default bench on this box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call14.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 --timeout-method=thread -k "long_stream" 2>&1 | tail -12 | tee -a $L
grep "full-size long" gpurun_out/gpu_ops_report.txt | tee -a $L
echo "== default bench ==" | tee -a $L
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json gpurun_out/_b_full.json > gpurun_out/_b.json 2> gpurun_out/_b.err ) 2>&1 | grep real | tee -a $L
python -c "
import json; d=json.loads(open('gpurun_out/_b.json').read().strip().splitlines()[-1]); print('value', d['value'], d['samples'], 'percall', d['one_frame_per_call']['value'], 'dropin', d['dropin_surface']['value'], 'roofline', d['roofline']['frac'], d['roofline']['mean_launch_ms'])" | tee -a $L
