#!/bin/bash
# round 4, call 12: bench.py --gpus 2 over gloo on one GPU after the partition is chosen with the MEASURED message time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call12.log
: > $L
n2() {
  local name=$1; shift
  echo "== N = 2 over gloo on one GPU: $name ==" | tee -a $L
  REFVSR_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 --repeats 1 --clip 20 --no-kernels --no-cpu-baseline --no-dropin "$@" 2> gpurun_out/_n2.err | tail -1 | cut -c1-14000 > gpurun_out/_n2.json
  python -c "
import json; d=json.load(open('gpurun_out/_n2.json')); w=d.get('wavefront') or {}
print('n2 value', round(d['value'],1), 'wavefront', {k: w.get(k) for k in ('ranks_seen','value','frames_equal','error')}, 'partition', (w.get('partition') or {}).get('name'), (w.get('partition') or {}).get('predicted_speedup'), w.get('partition_chosen_with'), 'ctx', {k: (w.get('context_exchange') or {}).get(k) for k in ('messages','bytes_per_message')}, 'handoff', {k: (w.get('handoff') or {}).get(k) for k in ('messages','ms_per_message_measured')})" 2>&1 | cut -c1-1200 | tee -a $L
  if ! grep -q '"value"' gpurun_out/_n2.json; then tail -8 gpurun_out/_n2.err | cut -c1-500 | tee -a $L; fi
}
n2 "default (exchange, partition by measured message time)"
cp gpurun_out/_n2.json gpurun_out/r04_bench_n2_gloo_one_gpu.json
n2 "exchange, block-cyclic 1 (every window imports two contexts, a hand-off per frame)" --wavefront-partition cyclic1
