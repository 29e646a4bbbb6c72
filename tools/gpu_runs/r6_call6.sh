#!/bin/bash
# round 6, call 6: one implementation of the P | F | M schedule (a steady call = a group of one window): the whole GPU suite + bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 500 python bench.py --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront > gpurun_out/r06_bench_call6.json 2> gpurun_out/r06_bench_call6.err; python -c "import json; d=json.load(open(\"gpurun_out/bench_full.json\")); print(d[\"value\"], d[\"samples\"], d[\"one_frame_per_call\"][\"value\"], d[\"dropin_surface\"][\"value\"], d[\"pcie_inclusive\"][\"value\"])"; tail -3 gpurun_out/r06_bench_call6.err
