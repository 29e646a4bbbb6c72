cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
echo "== default =="; python tools/bench_conv48.py 2>&1 | grep conv
echo "== NO_W16 =="; REFVSR_CONV_NO_W16=1 python tools/bench_conv48.py 2>&1 | grep conv
echo "== NO_NW8 NO_W16 (round-2 start) =="; REFVSR_CONV_NO_W16=1 REFVSR_CONV_NO_NW8=1 python tools/bench_conv48.py 2>&1 | grep conv
echo "== SQ counters: LR1080 48->48 and 2x2160 two-source =="
for c in "LR1080" "2x2160"; do
rm -rf gpurun_out/pmc48
(cd /tmp && CONV48_ONLY=$c CONV48_ITERS=4 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OLDPWD/gpurun_out/pmc48" -o k -- python "$OLDPWD/tools/bench_conv48.py" > /dev/null 2>&1)
(cd /tmp && CONV48_ONLY=$c CONV48_ITERS=4 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d "$OLDPWD/gpurun_out/pmc48b" -o k -- python "$OLDPWD/tools/bench_conv48.py" > /dev/null 2>&1)
python - <<'PY'
import csv, collections, glob
for d in ('gpurun_out/pmc48', 'gpurun_out/pmc48b'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    rows = list(csv.DictReader(open(fs[0])))
    acc = collections.OrderedDict()
    for r in rows:
        if 'conv_mfma' not in r['Kernel_Name']: continue
        k = (r['Kernel_Name'][:64], r.get('Grid_Size', ''))
        acc.setdefault(k, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, m in acc.items():
        print(k[0], k[1], ' '.join('%s=%.4g' % (c.replace('SQ_', ''), sum(v) / len(v)) for c, v in m.items()))
PY
rm -rf gpurun_out/pmc48 gpurun_out/pmc48b
done
