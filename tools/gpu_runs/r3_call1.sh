# round 3, call 1: first run of the compile-time-specialised 24-channel ResBlock kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2))'
echo "== resblock24 op test"; timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "resblock" 2>&1 | tail -15
echo "== microbench"; timeout 300 python tools/bench_resblock.py 2>&1 | grep resblock
echo "== microbench under rocprofv3 (device durations)"
( cd /tmp && RB_ITERS=3 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rb -o rb -- python $OLDPWD/tools/bench_resblock.py > /dev/null 2>&1; f=$(find /tmp/prof_rb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200 && cp "$f" $OLDPWD/gpurun_out/r3_call1_rb_kernel_stats.csv )
echo "== full gpu suite"; timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
echo "== bench (rb24)"; timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels 2>&1 | tail -1 | python -c "$fmt"
echo "== bench (generic lean)"; REFVSR_NO_RB24=1 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels 2>&1 | tail -1 | python -c "$fmt"
echo "== bench (rb24) again"; timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels 2>&1 | tail -1 | python -c "$fmt"
