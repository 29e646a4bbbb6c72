#!/bin/bash
# round 5, call 11: group mode A/Bs on one box -- two M streams alternating between groups, groups in flight, group size; the
# host-bound regime (64 x 96) with and without groups; kernel trace of the default mode (P | F | M, G = 4)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call11.log
: > $L
fmt='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value %.1f  samples %s  roofline %s  streams %s" % (d["value"], d["samples"], d["roofline"] and (round(d["roofline"].get("frac"),4), d["roofline"].get("mean_launch_ms")), d.get("streams_ms_per_frame")))'
B="python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs --full-json gpurun_out/_b_full.json"
run() {
  local name=$1; shift
  echo "== $name ==" | tee -a $L
  env "$@" timeout 240 $B $EXTRA > gpurun_out/_b.out 2> gpurun_out/_b.err
  tail -1 gpurun_out/_b.out | python -c "$fmt" 2>/dev/null | cut -c1-400 | tee -a $L || true
  if ! tail -1 gpurun_out/_b.out | grep -q '"value"'; then tail -4 gpurun_out/_b.err | cut -c1-400 | tee -a $L; fi
}
EXTRA=""
run "G4 default" X=1
run "G4 two M streams" REFVSR_GROUP_TWO_M=1
run "G4 pipe depth 4 (3 groups in flight)" REFVSR_PIPE_DEPTH=4
run "G4 pipe depth 2 (1 group in flight)" REFVSR_PIPE_DEPTH=2
run "G4 default (again)" X=1
EXTRA="--group 3"
run "G3" X=1
EXTRA="--size 64x96"
run "64x96 G4" X=1
EXTRA="--size 64x96 --group 1"
run "64x96 G1" X=1
echo "== kernel trace, default mode ==" | tee -a $L
rm -rf gpurun_out/prof_d
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_d" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof_d.log" 2>&1)
f=$(find gpurun_out/prof_d -name "*kernel_trace.csv" | head -1)
python tools/trace_analysis.py $f 8 20 > gpurun_out/r05_trace_analysis.txt 2>&1
head -8 gpurun_out/r05_trace_analysis.txt | tee -a $L
rm -rf gpurun_out/prof_d
