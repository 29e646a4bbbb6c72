#!/bin/bash
# round 6, call 2: CU-mask placement probe, the micro-benchmark at the frame's proportions, the frame-group bench under CU splits
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe > gpurun_out/r06_cu_mask_probe.txt 2>&1
cat gpurun_out/r06_cu_mask_probe.txt
timeout 300 python tools/bench_cu_partition.py > gpurun_out/r06_cu_partition_microbench.txt 2>&1
cat gpurun_out/r06_cu_partition_microbench.txt
B="python bench.py --no-other-configs --no-cpu-baseline --no-kernels --no-dropin"
for split in "" 128,32,96 144,32,80 160,32,64 128,64,64 176,32,48 ""; do
  echo "== REFVSR_CU_SPLIT='$split'" | tee -a gpurun_out/r06_cu_split_bench_ab.txt
  REFVSR_CU_SPLIT=$split timeout 200 $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d.get('config', {}).get('headline_mode'), d.get('streams_ms_per_frame'))" | tee -a gpurun_out/r06_cu_split_bench_ab.txt
done
