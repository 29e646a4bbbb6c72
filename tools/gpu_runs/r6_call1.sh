#!/bin/bash
# round 6, call 1: does the tree still pass on the GPU + the CU-partition micro-benchmark
set -x
mkdir -p gpurun_out
python -c "import torch; print(torch.cuda.get_device_name(0))"
timeout 300 python tools/bench_cu_partition.py > gpurun_out/r06_cu_partition_microbench.txt 2>&1
cat gpurun_out/r06_cu_partition_microbench.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py > gpurun_out/r06_bench_call1.json 2> gpurun_out/r06_bench_call1.err; tail -c 1500 gpurun_out/r06_bench_call1.json
