cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "resblock" 2>&1 | tail -30
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export REFVSR_NO_CHAIN_CALLS=1; else unset REFVSR_NO_CHAIN_CALLS; fi
  echo "host-bound 64x96: no_chain_calls=$v"
  timeout 300 python bench.py --size 64x96 --steps 60 --warmup 5 --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 | python -c "$fmt"
done
