#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_call10.log
: > $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2), "; match", round(d["roofline"]["mean_launch_ms"],3),"ms", round(d["roofline"]["frac"],3))'
echo "== op tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k resblock 2>&1 | tail -5 | tee -a $L
for k in 8 4 8 4; do
  echo "== bench REFVSR_RESBLOCK_WAVES=$k ==" | tee -a $L
  REFVSR_RESBLOCK_WAVES=$k timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
echo "== microbench under rocprof (8 waves) ==" | tee -a $L
rm -rf gpurun_out/prof_micro
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_micro" -o micro -- python "$OLDPWD/tools/bench_kernels.py" > "$OLDPWD/gpurun_out/microbench_rocprof.log" 2>&1)
python tools/trace_by_shape.py gpurun_out/prof_micro/micro_kernel_trace.csv 100 > gpurun_out/r2_micro_by_shape.txt 2>&1
grep -E "resblock|^kernel|resize" gpurun_out/r2_micro_by_shape.txt | cut -c1-170 | tee -a $L
rm -f gpurun_out/prof_micro/*.db
