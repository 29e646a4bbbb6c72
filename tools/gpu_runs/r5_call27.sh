#!/bin/bash
# round 5, call 27: P | F | M from the first stream on (C = 24): groups cut at the restarts vs fixed groups of four, same box; RefVSR_MFID leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r05_group_cut_ab.txt
echo "== after the layout fix (P | F | M from the first stream on) and pipelined restarts: fixed groups vs groups cut at the restarts ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -p no:cacheprovider -k "pipelined_mode or steady_windows" 2>&1 | tail -2 | tee -a $L
for rep in 1 2; do
for steps in 20 100; do
for fixed in 1 ""; do
  REFVSR_BENCH_FIXED_GROUPS=$fixed timeout 600 python bench.py --steps $steps --warmup 5 --repeats 3 --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront \
      --full-json gpurun_out/_gcut_full.json > gpurun_out/_gcut.json 2> gpurun_out/_gcut.err
  python - <<PY | tee -a $L
import json
j=json.load(open('gpurun_out/_gcut.json'))
print('rep $rep steps $steps fixed=[$fixed]: groups', round(j['value'],2), j['samples'], ' per-call', j['one_frame_per_call']['value'], ' dropin', j['dropin_surface']['value'], ' pcie', j['pcie_inclusive']['value'], j['config']['pipe_layout'])
PY
done
done
done
echo "== RefVSR_MFID ==" | tee -a $L
timeout 600 python bench.py --config config_RefVSR_MFID --no-other-configs --no-cpu-baseline --no-kernels --no-wavefront --group 1 --no-dropin \
    --steps 12 --warmup 4 --repeats 3 --full-json gpurun_out/_mf_full.json > gpurun_out/_mf.json 2> gpurun_out/_mf.err
python - <<'PY' | tee -a $L
import json
j=json.load(open('gpurun_out/_mf.json'))
print('RefVSR_MFID one frame per call:', round(j['value'],2), j['samples'], j['config']['pipe_layout'])
PY
