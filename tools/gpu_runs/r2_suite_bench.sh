cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2))'
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -2
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels 2>&1 | tail -1 | python -c "$fmt"
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels 2>&1 | tail -1 | python -c "$fmt"
