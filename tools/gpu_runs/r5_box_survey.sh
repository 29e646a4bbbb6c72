#!/bin/bash
# round 5: one bench process on a fresh box (called several times: the boxes of the pool differ by up to 8 %)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
fmt='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f  samples %s  percall %s dropin %s roofline %s" % (d["value"], d["samples"], d.get("one_frame_per_call") and round(d["one_frame_per_call"]["value"],1), d.get("dropin_surface") and round(d["dropin_surface"]["value"],1), d["roofline"] and (round(d["roofline"].get("frac"),4), d["roofline"].get("mean_launch_ms"))))'
( rocm-smi --showproductname 2>/dev/null | grep -i "card series\|GUID" | head -2; cat /proc/cpuinfo | grep "model name" | head -1 ) | tr '\n' ' '
echo
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs --full-json gpurun_out/_b.json | python -c "$fmt"
