#!/bin/bash
# round 6, call 3: phase-A groups -- the new parity tests, the N = 2 protocol run of configs[3] at its size, the N = 1 line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "phase_a_group or two_process or bench_gpus_2 or two_phase or context_exchange_block" 2>&1 | tail -15
cp /tmp/bench_n2_configs3_full.json gpurun_out/r06_bench_n2_configs3_full.json 2>/dev/null
timeout 400 python bench.py --no-other-configs --no-cpu-baseline --no-kernels > gpurun_out/r06_bench_call3.json 2> gpurun_out/r06_bench_call3.err; tail -c 2500 gpurun_out/r06_bench_call3.json; tail -5 gpurun_out/r06_bench_call3.err
