#!/bin/bash
# round 4, call 9: the context exchange of shard.run_wavefront on the GPU: (a) the engine-level tests (context round trip, two-process
# exchange over gloo on the one GPU); (b) the N = 2 protocol of bench.py over gloo with and without the exchange, restart-free and
# with restarts (frames_equal must hold); (c) the default bench's wavefront_model with the measured context terms.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call9.log
: > $L
echo "== tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 240 --timeout-method=thread -k "context_export or two_process or packed_state" 2>&1 | tail -8 | tee -a $L
n2() {
  local name=$1; shift
  echo "== N = 2 over gloo on one GPU: $name ==" | tee -a $L
  REFVSR_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 --repeats 1 --clip 20 --no-kernels --no-cpu-baseline --no-dropin "$@" 2> gpurun_out/_n2.err | tail -1 | cut -c1-12000 > gpurun_out/_n2.json
  python -c "
import json; d=json.load(open('gpurun_out/_n2.json')); w=d.get('wavefront') or {}
print('n2 value', round(d['value'],1), 'wavefront', {k: w.get(k) for k in ('ranks_seen','value','frames_equal','error')}, 'partition', (w.get('partition') or {}).get('name'), (w.get('partition') or {}).get('predicted_speedup'), 'ctx', {k: (w.get('context_exchange') or {}).get(k) for k in ('messages','bytes_per_message','host_seconds_blocked_waiting_all_ranks')}, 'handoff msgs', (w.get('handoff') or {}).get('messages'))" 2>&1 | cut -c1-900 | tee -a $L
  if ! grep -q '"value"' gpurun_out/_n2.json; then tail -5 gpurun_out/_n2.err | cut -c1-500 | tee -a $L; fi
}
n2 "exchange (default partition choice)"
cp gpurun_out/_n2.json gpurun_out/r04_bench_n2_gloo_one_gpu.json
n2 "no exchange" --no-wavefront-exchange
n2 "exchange, block-cyclic 2" --wavefront-partition cyclic2
n2 "exchange, growing block-cyclic" --wavefront-partition cyclic_growing
n2 "exchange, balanced (hand-off inside a restart unit)" --wavefront-partition balanced
echo "== default bench: wavefront_model ==" | tee -a $L
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-other-configs 2> gpurun_out/_b.err | tail -1 > gpurun_out/_b.json
python -c "
import json; d=json.load(open('gpurun_out/_b.json')); w=d['wavefront_model']
print('value', round(d['value'],1), d['samples'])
print('phases', w['phase_ms_per_frame_measured'])
print('exchange terms', {k: w['context_exchange'][k] for k in ('context_prepare_ms','cold_window_extra_with_contexts_ms','message_ms_assumed')})
for n in ('2','4','8'):
    e=w['predicted_speedup'][n]
    print(n, json.dumps(e)[:1500])
" 2>&1 | cut -c1-1800 | tee -a $L
if ! grep -q '"value"' gpurun_out/_b.json; then tail -5 gpurun_out/_b.err | cut -c1-500 | tee -a $L; fi
cp gpurun_out/_b.json gpurun_out/r04_bench_wavefront_model_exchange.json
