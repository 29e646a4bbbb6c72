#!/bin/bash
# round 4, call 13: RefVSR_IR on the two internal streams: bit-identity test, then the bench with and without
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call13.log
: > $L
echo "== tests ==" | tee -a $L
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 120 --timeout-method=thread -k "refvsr_ir" > gpurun_out/_t.out 2>&1
grep -i -B2 -A14 "Traceback\|^E " gpurun_out/_t.out | head -60 | cut -c1-300 | tee -a $L
tail -3 gpurun_out/_t.out | tee -a $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", d.get("samples"), "pipelined", d["config"]["pipelined_calls"])'
for round in 1 2; do
  echo "== RefVSR_IR_MFID pipelined (round $round) ==" | tee -a $L
  timeout 200 python bench.py --config config_RefVSR_IR_MFID --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin --no-other-configs 2> gpurun_out/_b.err | tail -1 > gpurun_out/r04_bench_IR_MFID.json
  python -c "$fmt" < gpurun_out/r04_bench_IR_MFID.json 2>&1 | tail -1 | tee -a $L
  if ! grep -q '"value"' gpurun_out/r04_bench_IR_MFID.json; then tail -5 gpurun_out/_b.err | cut -c1-400 | tee -a $L; fi
  echo "== RefVSR_IR_MFID --no-pipeline (round $round) ==" | tee -a $L
  timeout 200 python bench.py --config config_RefVSR_IR_MFID --steps 10 --warmup 2 --repeats 3 --no-pipeline --no-cpu-baseline --no-kernels --no-wavefront --no-dropin --no-other-configs 2>/dev/null | tail -1 | python -c "$fmt" 2>&1 | tail -1 | tee -a $L
done
