#!/bin/bash
# round 5, call 13: frame groups on the other configurations (HD matching, x2, 7-frame windows, C = 48 fall-back) + N = 2 under the
# driver's launch form on one GPU (backend chosen automatically)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call13.log
: > $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 400 --timeout-method=thread -k "frame_groups" 2>&1 | tail -8 | tee -a $L
echo "== driver launch form, N = 2 on one GPU ==" | tee -a $L
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 8 --warmup 5 --repeats 2 --no-kernels --size 64x96 --clip 16 --full-json gpurun_out/_n2.json 2> gpurun_out/_n2.err | tail -1 | cut -c1-1200 | tee -a $L
tail -3 gpurun_out/_n2.err | cut -c1-300 | tee -a $L
