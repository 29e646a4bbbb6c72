# round 3, call 8: conv48 workgroup shape A/B (16 waves x 2 groups vs 8 waves x 4 groups), 8K at size with the size guard
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2))'
echo "== 8K test + conv48 tests"; timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "8k or conv48" 2>&1 | tail -3
echo "== conv48 tests, 8 waves"; REFVSR_CONV48_WAVES=8 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "conv48" 2>&1 | tail -3
for i in 1 2; do
echo "== MFID (16 waves) $i"; timeout 400 python bench.py --config config_RefVSR_MFID --steps 20 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 | python -c "$fmt"
echo "== MFID (8 waves) $i"; REFVSR_CONV48_WAVES=8 timeout 400 python bench.py --config config_RefVSR_MFID --steps 20 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront 2>&1 | tail -1 | python -c "$fmt"
done
echo "== MFID_8K 1080p (16 waves)"; timeout 600 python bench.py --config config_RefVSR_MFID_8K --size 1080x1920 --frames 5 --steps 6 --warmup 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8K', round(d['value'],2), 'fps', round(d['ms_per_step'],1), 'ms')"
echo "== MFID_8K 1080p (8 waves)"; REFVSR_CONV48_WAVES=8 timeout 600 python bench.py --config config_RefVSR_MFID_8K --size 1080x1920 --frames 5 --steps 6 --warmup 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8K', round(d['value'],2), 'fps', round(d['ms_per_step'],1), 'ms')"
echo "== MFID_8K 1080p (generic)"; REFVSR_NO_CONV24=1 timeout 600 python bench.py --config config_RefVSR_MFID_8K --size 1080x1920 --frames 5 --steps 6 --warmup 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8K', round(d['value'],2), 'fps', round(d['ms_per_step'],1), 'ms')"
