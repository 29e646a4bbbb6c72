#!/bin/bash
# round 6, call 9: a caller's first frame on the three streams (a member of its group like a roll-over): the e2e suite + bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_evalrun.py tests/test_torch_ops.py -m gpu -x -q 2>&1 | tail -8
timeout 500 python bench.py --no-other-configs --no-cpu-baseline --no-kernels > gpurun_out/r06_bench_call9.json 2> gpurun_out/r06_bench_call9.err; python -c "
import json; d=json.load(open('gpurun_out/bench_full.json')); print(d['value'], d['samples'], d['one_frame_per_call']['value'], d['dropin_surface']['value'], d['first_frame_ms']); wm=d['wavefront_model']; print(wm['one_rank_same_clip'], wm['one_rank_wavefront'])"; tail -3 gpurun_out/r06_bench_call9.err
