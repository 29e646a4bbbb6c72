#!/bin/bash
# round 5, call 3: bench.py with frame groups as the headline mode (G = 4), the compact line, `--gpus 2` self-launch on one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call3.log
: > $L
echo "== bench self-launch test ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 500 --timeout-method=thread -x -k "bench_gpus_2" 2>&1 | tail -15 | tee -a $L
echo "== default bench ==" | tee -a $L
timeout 900 python bench.py --steps 20 --warmup 5 --full-json gpurun_out/r5_bench_full.json > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
echo "rc $? line bytes $(wc -c < gpurun_out/r5_bench.json)" | tee -a $L
tail -c 6000 gpurun_out/r5_bench.json | tee -a $L
tail -5 gpurun_out/r5_bench.err | tee -a $L
for g in 1 2 3; do
  echo "== bench --group $g ==" | tee -a $L
  timeout 300 python bench.py --steps 20 --warmup 5 --group $g --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --full-json gpurun_out/r5_bench_g$g.json 2>> gpurun_out/r5_bench.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.1f samples %s dropin %s percall %s roofline %s streams %s' % (d['value'], d['samples'], d.get('dropin_surface'), d.get('one_frame_per_call'), d['roofline'] and (d['roofline'].get('frac'), d['roofline'].get('mean_launch_ms'), d['roofline'].get('maps_per_launch')), d.get('streams_ms_per_frame')))" | tee -a $L
done
