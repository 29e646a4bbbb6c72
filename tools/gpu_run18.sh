#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/run18.log
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; match", round(d["roofline"]["mean_launch_ms"],3),"ms", round(d["roofline"]["frac"],3))'
echo "== bench priority ON ==" | tee $L
REFVSR_STREAM_PRIORITY=1 timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/bench_prio.log; python -c "$fmt" < gpurun_out/bench_prio.log | tee -a $L
echo "== bench priority OFF ==" | tee -a $L
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
echo "== rocprof priority ON ==" | tee -a $L
rm -rf gpurun_out/prof_prio
(cd /tmp && REFVSR_STREAM_PRIORITY=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_prio" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof_prio.log" 2>&1)
tail -1 gpurun_out/rocprof_prio.log | python -c "$fmt" | tee -a $L
grep match_top2 gpurun_out/prof_prio/bench_kernel_stats.csv | cut -c1-120 | tee -a $L
