#!/bin/bash
# RefVSR_MFID (C = 48): one vs two alternating M streams, 4 interleaved runs each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call33.log
: > $L
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2))'
B="python bench.py --config config_RefVSR_MFID --steps 20 --warmup 3 --no-cpu-baseline --no-kernels --no-wavefront --no-dropin"
for i in 1 2 3 4; do
a=$(timeout 300 $B 2>&1 | tail -1 | python -c "$fmt")
b=$(REFVSR_PIPE_TWO_M=1 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt")
echo "MFID one M $a   two M $b" | tee -a $L
done
