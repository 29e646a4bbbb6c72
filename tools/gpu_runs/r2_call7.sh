#!/bin/bash
# new bench (N=1 full JSON, N=2 over gloo on the one GPU), PMC passes of the listed kernels, parity tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r2_call7.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== pmc ==" | tee -a $L
rm -rf gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_fetch" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_k_write" -o k -- python "$OLDPWD/tools/pmc_kernels.py" > "$OLDPWD/gpurun_out/pmc_k_write.log" 2>&1)
python tools/pmc_to_json.py gpurun_out/pmc_k_fetch gpurun_out/pmc_k_write profiles 2>&1 | cut -c1-1500 | tee -a $L
cp profiles/pmc_kernels.json profiles/pmc_match_top2.json gpurun_out/ 2>/dev/null
echo "== bench N=1 ==" | tee -a $L
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r2.json; python -c "
import json; d=json.load(open('gpurun_out/bench_r2.json'))
print('value', round(d['value'],2), 'dropin', d['dropin_surface'] and round(d['dropin_surface']['value'],2), 'match frac', round(d['roofline']['frac'],3), 'first', d['first_frame_ms'])
print('whole', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['whole_path'].items() if k not in ('counting','breakdown')})
for k in d.get('kernels',[]): print('  %-52s %7.1f us  %7.1f TF (%.3f)  %7.0f GB/s (%.3f) traffic %s' % (k['kernel'],k['us_per_launch'],k['tflops'],k['frac_mfma'],k['gbs'],k['frac_hbm'],k['traffic']))
print('cpu', d['cpu_baseline'])
" 2>&1 | tee -a $L
echo "== bench N=2 (gloo, one GPU) ==" | tee -a $L
REFVSR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --clip 20 --no-kernels 2>&1 | tail -3 | cut -c1-3000 | tee gpurun_out/bench_n2_gloo.log | tee -a $L
echo "== tests ==" | tee -a $L
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -n 4 -x 2>&1 | tail -12 | tee -a $L
grep -E "recurrence|near-ties|full-size" gpurun_out/gpu_ops_report.txt | tee -a $L
