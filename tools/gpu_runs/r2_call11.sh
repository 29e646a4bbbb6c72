#!/bin/bash
# split-fp16 exact search: match tests, micro-benchmark of the matching path, A/B of the bench with margin 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_call11.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2), "; match", round(d["roofline"]["mean_launch_ms"],3),"ms", round(d["roofline"]["frac"],3))'
echo "== match tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_torch_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "match or torch" 2>&1 | tail -8 | tee -a $L
grep -i match gpurun_out/gpu_ops_report.txt | tee -a $L
echo "== matching micro-benchmark ==" | tee -a $L
timeout 200 python - <<'P' 2>&1 | tail -12 | tee -a $L
import sys; sys.path.insert(0, 'tools')
import bench_kernels as bk
bk.bench_match()
P
for m in default 0 default 0; do
  echo "== bench margin $m ==" | tee -a $L
  if [ $m = default ]; then a=""; else a="--match-margin $m"; fi
  timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin $a 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
echo "== full GPU suite ==" | tee -a $L
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x --durations=4 2>&1 | tail -12 | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r2_call11_parity_report.txt 2>/dev/null
