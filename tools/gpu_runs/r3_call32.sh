#!/bin/bash
# run-to-run spread of the bench with one and with two alternating M streams (8 runs each, interleaved)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call32.log
: > $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2))'
B="python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin"
for i in 1 2 3 4 5 6 7 8; do
a=$(timeout 300 $B 2>&1 | tail -1 | python -c "$fmt")
b=$(REFVSR_PIPE_TWO_M=1 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt")
echo "one M $a   two M $b" | tee -a $L
done
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x -k "pipelined or two_phase or deterministic" 2>&1 | tail -3 | tee -a $L
