#!/bin/bash
# round 4, call 4: (a) the N = 2 protocol of bench.py on one GPU over gloo (new wavefront leg: measured phases -> chosen
# partition -> two-lane run -> checksums), (b) finer A/B of the backward head, (c) the fused 48-channel block: bit-identity
# tests, microbench, RefVSR_MFID / MFID_8K frame rates with and without it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call4.log
: > $L
echo "== N = 2 protocol, two ranks on one GPU over gloo ==" | tee -a $L
REFVSR_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --repeats 2 --clip 20 --no-kernels --no-cpu-baseline > gpurun_out/_n2.out 2> gpurun_out/_n2.err
tail -1 gpurun_out/_n2.out | cut -c1-9000 > gpurun_out/r04_bench_n2_gloo_one_gpu.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_n2_gloo_one_gpu.json')); w=d.get('wavefront')
print('n2 value', d['value'], d.get('samples'))
print('wavefront', {k: w.get(k) for k in ('ranks_seen','backend','value','frames_equal','partition','handoff','speedup_over_one_rank_phase_sum')} if w else None)" 2>&1 | cut -c1-1500 | tee -a $L
tail -3 gpurun_out/_n2.err | cut -c1-300 | tee -a $L
echo "== N = 2, restart-free regime forced to the block-cyclic partition ==" | tee -a $L
REFVSR_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 6 --warmup 2 --repeats 1 --clip 20 --wavefront-partition cyclic3 --no-kernels --no-cpu-baseline --no-dropin 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); w=d.get('wavefront')
print('cyclic3', {k: w.get(k) for k in ('value','frames_equal','partition','handoff')} if w else None)" 2>&1 | cut -c1-900 | tee -a $L
fmt='import sys,json
d=json.loads(sys.stdin.read())
print("value %.1f  samples %s  dropin %s" % (d["value"], d["samples"], d["dropin_surface"] and round(d["dropin_surface"]["value"],1)))'
B="python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs"
run() {
  local name=$1; shift
  echo "== $name ==" | tee -a $L
  env "$@" timeout 240 $B $EXTRA > gpurun_out/_b.out 2> gpurun_out/_b.err
  tail -1 gpurun_out/_b.out | python -c "$fmt" 2>/dev/null | cut -c1-300 | tee -a $L || true
  if ! tail -1 gpurun_out/_b.out | grep -q '"value"'; then tail -4 gpurun_out/_b.err | cut -c1-400 | tee -a $L; fi
}
EXTRA=""
for round in 1 2; do
  run "REFVSR_BW_HEAD_BLOCKS=8 (round $round)" REFVSR_BW_HEAD_BLOCKS=8
  run "default = 12 (round $round)" X=1
  run "REFVSR_BW_HEAD_BLOCKS=16 (round $round)" REFVSR_BW_HEAD_BLOCKS=16
done
echo "== resblock48: bit-identity tests ==" | tee -a $L
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider --timeout 100 --timeout-method=thread -k "resblock48" 2>&1 | tail -12 | tee -a $L
echo "== resblock48: microbench ==" | tee -a $L
timeout 200 python tools/bench_resblock48.py 2>&1 | grep resblock48 | tee gpurun_out/r04_resblock48_microbench.txt | tee -a $L
echo "== MFID end-to-end tests (fixtures, live oracle, full-size reference fixture, equivalence with the two-launch path) ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 240 --timeout-method=thread -k "MFID or HD48 or F_16x24 or round4 or 8k" 2>&1 | tail -8 | tee -a $L
EXTRA="--config config_RefVSR_MFID --steps 12 --warmup 3"
for round in 1 2; do
  run "RefVSR_MFID default (round $round)" X=1
  run "RefVSR_MFID REFVSR_NO_RB48=1 (round $round)" REFVSR_NO_RB48=1
done
echo "== RefVSR_MFID_8K 1080p -> 8K (other_configs leg of the default bench), with and without the fused block ==" | tee -a $L
for K in X REFVSR_NO_RB48; do
  env $K=1 timeout 300 python -c "
import bench, torch, json
r = bench.other_config_leg('config_RefVSR_MFID_8K', 1080, 1920, 4, 2, torch.device('cuda:0'), repeats=2)
print('$K', round(r['value'], 3), r['samples'], r['peak_memory_gib'])" 2>&1 | tail -1 | tee -a $L
done
