#!/bin/bash
# match_top2: max3 reductions (-fno-honor-nans TU), partner-lane threshold, unmasked inserts; coalesced match_patches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_call13.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2), "; match", round(d["roofline"]["mean_launch_ms"],3),"ms", round(d["roofline"]["frac"],3))'
echo "== match tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_torch_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "match or torch" 2>&1 | tail -8 | tee -a $L
grep -i "match" gpurun_out/gpu_ops_report.txt | tee -a $L
echo "== matching micro-benchmark ==" | tee -a $L
timeout 200 python - <<'P' 2>&1 | tail -12 | tee -a $L
import sys; sys.path.insert(0, 'tools')
import bench_kernels as bk
bk.bench_match()
P
for i in 1 2; do
  timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
echo "== stream tests ==" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4 | tee -a $L
echo "== SQ instruction counters ==" | tee -a $L
rm -rf gpurun_out/pmc_sq2
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d "$OLDPWD/gpurun_out/pmc_sq2" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_sq2.log" 2>&1)
python - <<'PY' 2>&1 | tee -a $L
import csv, collections, glob
fs = glob.glob('gpurun_out/pmc_sq2/**/*counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(fs[0])))
acc = collections.OrderedDict()
for r in rows:
    k = (r['Kernel_Name'][:60], r.get('Grid_Size', ''))
    acc.setdefault(k, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
for k, m in acc.items():
    if 'match_top2' in k[0]:
        print(k[0][:48], k[1], ' '.join('%s=%.4g' % (c.replace('SQ_', ''), sum(v) / len(v)) for c, v in m.items()))
PY
find gpurun_out/pmc_sq2 -name "*counter_collection.csv" -exec cp {} gpurun_out/r2_pmc_sq_insts_v5.csv \;
rm -rf gpurun_out/pmc_sq2
