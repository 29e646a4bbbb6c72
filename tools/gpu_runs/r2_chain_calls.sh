cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2), "first", d.get("first_frame_ms"))'
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "resblock" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -2
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export REFVSR_NO_CHAIN_CALLS=1; else unset REFVSR_NO_CHAIN_CALLS; fi
  echo "no_chain_calls=$v"
  timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels 2>&1 | tail -1 | python -c "$fmt"
done
unset REFVSR_NO_CHAIN_CALLS
python tools/host_profile.py 2>&1 | grep "host enqueue"
REFVSR_NO_CHAIN_CALLS=1 python tools/host_profile.py 2>&1 | grep "host enqueue"
