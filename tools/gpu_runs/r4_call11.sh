#!/bin/bash
# round 4, call 11: the context exchange with engine-level ids (call 9 named the contexts f where the engine's windows say (0, f): every
# context was silently prepared again -- found by the test's "prepared exactly the own frames" count; strict mode now raises):
# the three tests, the N = 2 protocol over gloo with the exchange, the default bench's wavefront_model on a quiet box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call11.log
: > $L
echo "== tests ==" | tee -a $L
timeout 500 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 120 --timeout-method=thread -k "context_export or exchange" > gpurun_out/_t.out 2>&1
grep -i -A12 "Traceback" gpurun_out/_t.out | head -60 | cut -c1-300 | tee -a $L
tail -4 gpurun_out/_t.out | tee -a $L
echo "== N = 2 over gloo on one GPU, context exchange ==" | tee -a $L
REFVSR_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 --repeats 1 --clip 20 --no-kernels --no-cpu-baseline --no-dropin 2> gpurun_out/_n2.err | tail -1 | cut -c1-12000 > gpurun_out/r04_bench_n2_gloo_one_gpu.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_n2_gloo_one_gpu.json')); w=d.get('wavefront') or {}
print('n2 value', round(d['value'],1), 'wavefront', {k: w.get(k) for k in ('ranks_seen','value','frames_equal','error')}, 'partition', (w.get('partition') or {}).get('name'), (w.get('partition') or {}).get('predicted_speedup'), 'ctx', {k: (w.get('context_exchange') or {}).get(k) for k in ('messages','bytes_per_message','host_seconds_blocked_waiting_all_ranks')}, 'handoff msgs', (w.get('handoff') or {}).get('messages'))
print('phases', w.get('phase_ms_per_frame_measured'))" 2>&1 | cut -c1-1200 | tee -a $L
if ! grep -q '"value"' gpurun_out/r04_bench_n2_gloo_one_gpu.json; then tail -8 gpurun_out/_n2.err | cut -c1-500 | tee -a $L; fi
echo "== default bench: wavefront_model ==" | tee -a $L
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-other-configs --no-dropin 2> gpurun_out/_b.err | tail -1 > gpurun_out/_b.json
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
