#!/bin/bash
# round 6, call 7: the new masked-stream test; 600-frame soak of the three call modes on the refactored engine; PCIe-inclusive legs vs hardware queues
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "cu_masked" 2>&1 | tail -3
timeout 900 python tools/soak.py --frames 600 > gpurun_out/r06_soak_600_frames.txt 2>&1; tail -12 gpurun_out/r06_soak_600_frames.txt
for q in "" 6 8; do
  echo "== GPU_MAX_HW_QUEUES='$q'" | tee -a gpurun_out/r06_pcie_hw_queues_ab.txt
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront --full-json gpurun_out/_q.json > /dev/null 2> gpurun_out/_q.err
  python -c "
import json
d = json.load(open('gpurun_out/_q.json')); p = d['pcie_inclusive']
print('resident', round(d['value'], 1), 'per-call', round(d['one_frame_per_call']['value'], 1), 'dropin', round(d['dropin_surface']['value'], 1), '| pcie fp32', round(p['value'], 1), 'uint8', round(p['result_uint8']['value'], 1), 'uint8 + frames once', round(p['result_uint8_frames_once']['value'], 1))" | tee -a gpurun_out/r06_pcie_hw_queues_ab.txt
done
