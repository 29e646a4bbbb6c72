#!/bin/bash
# final defaults (M streams by model width): pipelined / stream tests, bench lines of RefVSR_small, RefVSR_MFID, MFID_8K
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call34.log
: > $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x -k "pipelined or two_phase or deterministic or stream_against" 2>&1 | tail -3 | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2))'
timeout 300 python bench.py --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
timeout 400 python bench.py --config config_RefVSR_MFID --steps 20 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 > gpurun_out/r03_bench_MFID.json
python -c "$fmt" < gpurun_out/r03_bench_MFID.json | tee -a $L
timeout 600 python bench.py --config config_RefVSR_MFID_8K --size 1080x1920 --frames 5 --steps 6 --warmup 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront 2>&1 | tail -1 > gpurun_out/r03_bench_MFID_8K_1080p.json
python -c "import json; d=json.load(open('gpurun_out/r03_bench_MFID_8K_1080p.json')); print('8K', round(d['value'],2), 'fps', round(d['ms_per_step'],1), 'ms')" | tee -a $L
