#!/bin/bash
# round 5, call 18: the 48-channel multi-map block (ABI 12) + frame groups for mid_channels = 48, and the PCIe-inclusive leg of bench.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r05_call18.log
: > $L
echo "== multi-map op tests + frame groups of the other configurations ==" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q -m gpu -p no:cacheprovider -k "multimap or frame_groups_other" 2>&1 | tail -6 | tee -a $L
echo "== resblock48 microbench ==" | tee -a $L
RB48_ONLY=1 timeout 300 python tools/bench_multimap.py 2>&1 | tee -a $L
for knob in 0 1; do
  echo "== RefVSR_MFID bench, REFVSR_NO_RB48_MULTIMAP=$knob ==" | tee -a $L
  REFVSR_NO_RB48_MULTIMAP=$knob timeout 600 python bench.py --config config_RefVSR_MFID --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront \
      --steps 12 --warmup 4 --repeats 3 --full-json gpurun_out/r05_bench_mfid_full_$knob.json > gpurun_out/r05_bench_mfid_$knob.json 2> gpurun_out/_m.err
  tail -c 3000 gpurun_out/_m.err | tail -3 | tee -a $L
  python - <<PY | tee -a $L
import json
j=json.load(open('gpurun_out/r05_bench_mfid_$knob.json'))
print('value', j['value'], 'samples', j.get('samples'), 'mode', j['config'].get('headline_mode'), 'G', j['config'].get('frames_per_call'))
print('one_frame_per_call', j.get('one_frame_per_call'), 'dropin', j.get('dropin_surface'), 'pcie', j.get('pcie_inclusive'))
print('layout', j['config'].get('pipe_layout'), 'streams', j.get('streams_ms_per_frame'))
PY
done
echo "== default bench (with the PCIe-inclusive leg) ==" | tee -a $L
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_call18.json 2> gpurun_out/_b.err ) 2>&1 | grep real | tee -a $L
tail -3 gpurun_out/_b.err | tee -a $L
cat gpurun_out/r05_bench_call18.json | tee -a $L
