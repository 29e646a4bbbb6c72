#!/bin/bash
# sequential GPU suite with per-test durations, torch ops, 8K at size + configs[4] / configs[2] bench lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_call8.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== new tests first ==" | tee -a $L
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x --durations=8 -k "torch_library or 8k_single or HD48 or long_recurrence" 2>&1 | tail -25 | tee -a $L
echo "== bench configs[4]: MFID_8K 1080x1920 ==" | tee -a $L
timeout 600 python bench.py --config config_RefVSR_MFID_8K --size 1080x1920 --frames 5 --steps 4 --warmup 2 --no-cpu-baseline --no-kernels 2>&1 | tail -1 | tee gpurun_out/bench_8k.json | cut -c1-1500 | tee -a $L
echo "== bench configs[2]: MFID 270x480 ==" | tee -a $L
timeout 600 python bench.py --config config_RefVSR_MFID --steps 12 --warmup 3 --no-cpu-baseline --no-kernels 2>&1 | tail -1 | tee gpurun_out/bench_mfid.json | cut -c1-900 | tee -a $L
echo "== full suite, sequential ==" | tee -a $L
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x --durations=15 --deselect tests/test_gpu_e2e.py::test_8k_single_window_at_size --deselect tests/test_gpu_e2e.py::test_long_recurrence_against_live_oracle 2>&1 | tail -30 | tee -a $L
grep -E "8K|HD48|recurrence" gpurun_out/gpu_ops_report.txt | tee -a $L
