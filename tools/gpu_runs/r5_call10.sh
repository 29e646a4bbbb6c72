#!/bin/bash
# round 5, call 10: resblock24 with two x-tile buffers in the sixteen-wave shape (next tile parked under conv2, three barriers per
# tile): bit-identity (8 vs 16 waves, store modes, multi-map, the fused output tail), micro-benchmarks, probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call10.log
: > $L
echo "== op tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider --timeout 200 --timeout-method=thread -x \
  -k "multimap or resblock24 or conv_hr_last or conv_stacks or compute_up" 2>&1 | tail -6 | tee -a $L
echo "== multimap microbench ==" | tee -a $L
timeout 300 python tools/bench_multimap.py 2>&1 | grep "^multimap resblock24" | tee -a $L
echo "== resblock microbench ==" | tee -a $L
timeout 300 python tools/bench_resblock.py 2>&1 | grep "rb24 default\|rb24 8 waves \|rb24 16 waves" | tee -a $L
echo "== probe (16 waves) ==" | tee -a $L
PROBE_WAVES=16 timeout 200 python tools/probe_resblock24.py 2>&1 | grep -A13 "2x 2nd tile" | tee -a $L
echo "== engine tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 300 --timeout-method=thread -x -k "frame_groups or stream_against_reference or full_size_against" 2>&1 | tail -4 | tee -a $L
echo "== bench ==" | tee -a $L
timeout 300 python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --full-json gpurun_out/_b.json | cut -c1-1500 | tee -a $L
