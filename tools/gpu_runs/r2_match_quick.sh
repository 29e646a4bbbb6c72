cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_torch_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "match or torch" 2>&1 | tail -3
timeout 200 python - <<'P' 2>&1 | tail -9
import sys; sys.path.insert(0, 'tools')
import bench_kernels as bk
bk.bench_match()
P
