#!/bin/bash
# quick regression: whole GPU suite sequentially + default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_quick.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x --durations=6 2>&1 | tail -25 | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2), "; match", round(d["roofline"]["mean_launch_ms"],3),"ms", round(d["roofline"]["frac"],3))'
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
