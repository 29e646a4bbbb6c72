cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export REFVSR_CONV_FORCE_NW8=1; else unset REFVSR_CONV_FORCE_NW8; fi
  echo "force_nw8=$v"; timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 | python -c "$fmt"
done
