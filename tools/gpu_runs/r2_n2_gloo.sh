#!/bin/bash
# the --gpus 2 protocol (headline + wavefront leg) with two ranks sharing ONE GPU over gloo: functional check only
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
REFVSR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --clip 20 --no-kernels --no-cpu-baseline 2>&1 | tail -2 | cut -c1-4000 | tee gpurun_out/r02_bench_n2_gloo_one_gpu.json
