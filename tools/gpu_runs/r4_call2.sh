#!/bin/bash
# round 4, call 2: the launches removed from a frame (refvsr_conf_alpha, refvsr_warp_nhwc16_up2, batched SPyNet pass, cached zero
# maps) and the deeper host run-ahead: (a) full GPU suite incl. the new bit-identity tests, (b) A/B of every knob through the
# repeated bench passes, (c) kernel trace -> launches per frame
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call2.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== new op tests ==" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "conf_alpha or warp_up2 or batched_conv" 2>&1 | tail -15 | tee -a $L
echo "== full suite ==" | tee -a $L
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=5 2>&1 | tail -25 | tee -a $L
fmt='import sys,json
d=json.loads(sys.stdin.read())
print("value %.1f  samples %s  dropin %.1f  M %.2f P %.2f F %.2f ms/call" % (d["value"], d["samples"], d["dropin_surface"]["value"], d["streams"]["median_pass"]["M_ms_per_call"], d["streams"]["median_pass"]["P_ms_per_call"], d["streams"]["median_pass"]["F_ms_per_call"]))'
B="python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernels --no-wavefront --no-other-configs"
for round in 1 2; do
  for K in default REFVSR_NO_FUSE_CONF REFVSR_NO_SPYNET_BATCH REFVSR_NO_WARP_UP2 REFVSR_PIPE_DEPTH2 ALL_OFF; do
    echo "== $K (round $round) ==" | tee -a $L
    case $K in
      default) timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" 2>&1 | cut -c1-300 | tee -a $L;;
      REFVSR_PIPE_DEPTH2) REFVSR_PIPE_DEPTH=2 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" 2>&1 | cut -c1-300 | tee -a $L;;
      ALL_OFF) REFVSR_NO_FUSE_CONF=1 REFVSR_NO_SPYNET_BATCH=1 REFVSR_NO_WARP_UP2=1 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" 2>&1 | cut -c1-300 | tee -a $L;;
      *) env $K=1 timeout 300 $B 2>&1 | tail -1 | python -c "$fmt" 2>&1 | cut -c1-300 | tee -a $L;;
    esac
  done
done
echo "== spynet microbench (one flow; two flows batched) ==" | tee -a $L
timeout 200 python tools/bench_spynet.py 2>&1 | grep "spynet" | tee gpurun_out/r04_spynet_microbench.txt | tee -a $L
echo "== rocprof kernel trace of the default bench ==" | tee -a $L
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-kernels --no-dropin --no-wavefront --no-other-configs > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200 | tee -a $L
python tools/trace_analysis.py gpurun_out/prof/bench_kernel_trace.csv 8 20 > gpurun_out/r04_call2_trace_analysis.txt 2>&1
head -34 gpurun_out/r04_call2_trace_analysis.txt | tee -a $L
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r04_call2_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof
