#!/bin/bash
# round 6, call 5: result formats (ABI 14), both roofs in the line, the full GPU suite on the round-6 tree
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 500 python bench.py --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront > gpurun_out/r06_bench_call5.json 2> gpurun_out/r06_bench_call5.err; python -c "import json; d=json.load(open(\"gpurun_out/bench_full.json\")); print(d[\"value\"], d[\"pcie_inclusive\"])"
