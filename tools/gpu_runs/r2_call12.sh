#!/bin/bash
# split-fp16 exact search v2 (B operand from the LR rows, LDS-direct double-buffered stages) + SQ counter breakdown
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_call12.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2),"fps", round(d["ms_per_step"],3),"ms; dropin", d["dropin_surface"] and round(d["dropin_surface"]["value"],2), "; match", round(d["roofline"]["mean_launch_ms"],3),"ms", round(d["roofline"]["frac"],3))'
echo "== match tests ==" | tee -a $L
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_torch_ops.py -m gpu -q --no-header -p no:cacheprovider -x -k "match or torch" 2>&1 | tail -8 | tee -a $L
grep -i "exact\|near" gpurun_out/gpu_ops_report.txt | tee -a $L
echo "== matching micro-benchmark ==" | tee -a $L
timeout 200 python - <<'P' 2>&1 | tail -12 | tee -a $L
import sys; sys.path.insert(0, 'tools')
import bench_kernels as bk
bk.bench_match()
P
for m in default 0 default 0; do
  echo "== bench margin $m ==" | tee -a $L
  if [ $m = default ]; then a=""; else a="--match-margin $m"; fi
  timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernels --no-dropin $a 2>&1 | tail -1 | python -c "$fmt" | tee -a $L
done
echo "== stream tests ==" | tee -a $L
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4 | tee -a $L
echo "== SQ counters (match_top2, resblock, convs) ==" | tee -a $L
rm -rf gpurun_out/pmc_sq
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OLDPWD/gpurun_out/pmc_sq" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_sq.log" 2>&1)
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$OLDPWD/gpurun_out/pmc_sq2" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_sq2.log" 2>&1)
python - <<'PY' 2>&1 | tee -a $L
import csv, collections, glob
for d in ('gpurun_out/pmc_sq', 'gpurun_out/pmc_sq2'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(d, 'no counter file'); continue
    rows = list(csv.DictReader(open(fs[0])))
    acc = collections.OrderedDict()
    for r in rows:
        k = (r['Kernel_Name'][:60], r.get('Grid_Size', ''), r.get('LDS_Block_Size', ''))
        acc.setdefault(k, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, m in acc.items():
        if not any(s in k[0] for s in ('match_top2', 'resblock', 'conv_mfma', 'warp_nhwc', 'match_exact')):
            continue
        mm = {c: sum(v) / len(v) for c, v in m.items()}
        print(k[0][:48], k[1], ' '.join('%s=%.4g' % (c.replace('SQ_', ''), v) for c, v in mm.items()))
PY
cp gpurun_out/pmc_sq/*/*counter_collection.csv gpurun_out/r2_pmc_sq_counters.csv 2>/dev/null || find gpurun_out/pmc_sq -name "*counter_collection.csv" -exec cp {} gpurun_out/r2_pmc_sq_counters.csv \;
find gpurun_out/pmc_sq2 -name "*counter_collection.csv" -exec cp {} gpurun_out/r2_pmc_sq_insts.csv \;
rm -rf gpurun_out/pmc_sq/*/*.db gpurun_out/pmc_sq2/*/*.db
