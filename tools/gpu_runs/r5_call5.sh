#!/bin/bash
# round 5, call 5: the group mode shrinks the kernel time (5.49 -> 5.08 ms per frame under rocprofv3) but not the frame: the P + F
# stream is the critical path (P 2.9 + F 1.9 = wall 4.8 ms per frame).  A/B: stream layouts, where the backward head runs, the
# workgroup shape of the multi-map launches (sixteen-wave workgroups own a CU's whole register file: nothing co-resides)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call5.log
: > $L
fmt='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f  samples %s  roofline %s  streams %s" % (d["value"], d["samples"], d["roofline"] and (round(d["roofline"].get("frac"),4), d["roofline"].get("mean_launch_ms")), d.get("streams_ms_per_frame")))'
B="python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs --full-json gpurun_out/_b_full.json"
run() {
  local name=$1; shift
  echo "== $name ==" | tee -a $L
  local g=4; for kv in "$@"; do case $kv in BENCH_GROUP=*) g=${kv#BENCH_GROUP=};; esac; done
  env "$@" timeout 240 $B --group $g > gpurun_out/_b.out 2> gpurun_out/_b.err
  tail -1 gpurun_out/_b.out | python -c "$fmt" 2>/dev/null | cut -c1-400 | tee -a $L || true
  if ! tail -1 gpurun_out/_b.out | grep -q '"value"'; then tail -4 gpurun_out/_b.err | cut -c1-400 | tee -a $L; fi
}
run "G4 default" X=1
run "G4 rb24 8 waves" REFVSR_RESBLOCK24_WAVES=8
run "G4 bw head off" REFVSR_BW_HEAD_BLOCKS=-1
run "G4 bw head off, 8 waves" REFVSR_BW_HEAD_BLOCKS=-1 REFVSR_RESBLOCK24_WAVES=8
run "G4 layout pfm" REFVSR_PIPE_LAYOUT=pfm
run "G4 layout pfm, bw head off" REFVSR_PIPE_LAYOUT=pfm REFVSR_BW_HEAD_BLOCKS=-1
run "G4 layout pfm, bw head off, 8 waves" REFVSR_PIPE_LAYOUT=pfm REFVSR_BW_HEAD_BLOCKS=-1 REFVSR_RESBLOCK24_WAVES=8
run "G4 layout p_fm" REFVSR_PIPE_LAYOUT=p_fm
run "G4 layout p_fm, 8 waves" REFVSR_PIPE_LAYOUT=p_fm REFVSR_RESBLOCK24_WAVES=8
run "G2 layout pfm, bw head off" REFVSR_PIPE_LAYOUT=pfm REFVSR_BW_HEAD_BLOCKS=-1 BENCH_GROUP=2
