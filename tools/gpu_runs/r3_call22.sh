#!/bin/bash
# bench.py after the input-only clip change: default line, N = 2 protocol over gloo on one GPU, smoke
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r3_call22.log
: > $L
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee -a $L
REFVSR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --clip 20 --no-kernels --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 | tee -a $L
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $L
