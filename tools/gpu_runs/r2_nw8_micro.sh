cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
echo "== default (8 waves) =="; CONV48_ONLY="S " CONV48_ITERS=100 python tools/bench_conv48.py 2>&1 | grep conv | cut -c1-70
echo "== NO_NW8 (4 waves) =="; CONV48_ONLY="S " CONV48_ITERS=100 REFVSR_CONV_NO_NW8=1 python tools/bench_conv48.py 2>&1 | grep conv | cut -c1-70
echo "== default (8 waves) =="; CONV48_ONLY="S " CONV48_ITERS=100 python tools/bench_conv48.py 2>&1 | grep conv | cut -c1-70
echo "== NO_NW8 (4 waves) =="; CONV48_ONLY="S " CONV48_ITERS=100 REFVSR_CONV_NO_NW8=1 python tools/bench_conv48.py 2>&1 | grep conv | cut -c1-70
