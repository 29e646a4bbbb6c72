cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms")'
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "conv" 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 | python -c "$fmt"; done
