#!/bin/bash
# after the 16-wave default: full GPU suite + benches of configs[1], [2], [4]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_after_w16.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
timeout 300 python bench.py --config config_RefVSR_MFID --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r02_bench_MFID.json
python -c "$fmt" < gpurun_out/r02_bench_MFID.json | tee -a $L
timeout 600 python bench.py --config config_RefVSR_MFID_8K --size 1080x1920 --frames 5 --steps 6 --warmup 2 --no-cpu-baseline --no-kernels 2>&1 | tail -1 > gpurun_out/r02_bench_MFID_8K_1080p.json
python -c "$fmt" < gpurun_out/r02_bench_MFID_8K_1080p.json | tee -a $L
