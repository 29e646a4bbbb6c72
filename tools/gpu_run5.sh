#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest gpu ==" | tee gpurun_out/run5.log
timeout 900 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider -n 2 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "^(FAILED|ERROR|SKIPPED)|passed|failed" gpurun_out/pytest_gpu_full.log | tail -30 | tee -a gpurun_out/run5.log
grep -E "^E  " gpurun_out/pytest_gpu_full.log | head -40 | tee -a gpurun_out/run5.log
echo "== bench ==" | tee -a gpurun_out/run5.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee gpurun_out/bench.log
echo "== bench N=2 code path (gloo, both ranks on GPU 0) ==" | tee -a gpurun_out/run5.log
REFVSR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 2>&1 | tail -2 | cut -c1-300 | tee gpurun_out/bench_n2_gloo.log
echo "== pmc conv ==" | tee -a gpurun_out/run5.log
rm -rf gpurun_out/pmc_conv1 gpurun_out/pmc_conv2
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OLDPWD/gpurun_out/pmc_conv1" -o c -- python "$OLDPWD/tools/pmc_conv.py" > "$OLDPWD/gpurun_out/pmc_conv1.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d "$OLDPWD/gpurun_out/pmc_conv2" -o c -- python "$OLDPWD/tools/pmc_conv.py" > "$OLDPWD/gpurun_out/pmc_conv2.log" 2>&1)
tail -3 gpurun_out/pmc_conv1.log | cut -c1-200; tail -3 gpurun_out/pmc_conv2.log | cut -c1-200
python - <<'PY'
import csv, glob, collections
for d in ('gpurun_out/pmc_conv1', 'gpurun_out/pmc_conv2'):
    for f in glob.glob(d + '/*counter_collection.csv'):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]
            if 'conv_mfma' in k or 'resblock' in k:
                acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in sorted(acc.items()):
            print('%-42s %-28s mean %.4g (n=%d)' % (k, c, sum(v) / len(v), len(v)))
PY
