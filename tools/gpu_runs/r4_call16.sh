#!/bin/bash
# round 4, call 16 (the last GPU minutes): the hand-off in two messages (EngineExecutor.split_handoff, default on) -- every
# two-process test with the real engine + the packed-state / context round trips, then the N = 2 protocol of bench.py over gloo
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call16.log
: > $L
echo "== tests ==" | tee -a $L
timeout 170 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 60 --timeout-method=thread -x -k "two_process or packed_state or context_export or ir_state_handoff" > gpurun_out/_t.out 2>&1
grep -i -A14 "Traceback" gpurun_out/_t.out | head -50 | cut -c1-300 | tee -a $L
tail -3 gpurun_out/_t.out | tee -a $L
echo "== N = 2 over gloo on one GPU ==" | tee -a $L
REFVSR_DIST_BACKEND=gloo timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --repeats 1 --warm-seconds 0.1 --clip 16 --no-kernels --no-cpu-baseline --no-dropin 2> gpurun_out/_n2.err | tail -1 | cut -c1-14000 > gpurun_out/_n2.json
python -c "
import json; d=json.load(open('gpurun_out/_n2.json')); w=d.get('wavefront') or {}
print('n2 value', round(d['value'],1), 'wavefront', {k: w.get(k) for k in ('ranks_seen','value','frames_equal','error')}, 'partition', (w.get('partition') or {}).get('name'), w.get('partition_chosen_with'), 'handoff', {k: str((w.get('handoff') or {}).get(k))[:160] for k in ('messages','format')})" 2>&1 | cut -c1-1200 | tee -a $L
if ! grep -q '"value"' gpurun_out/_n2.json; then tail -8 gpurun_out/_n2.err | cut -c1-500 | tee -a $L; fi
