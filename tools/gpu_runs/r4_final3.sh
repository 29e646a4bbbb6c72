#!/bin/bash
# round 4, last run of the final tree: full GPU suite, smoke, default bench (-> profiles/r04_bench.json)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_final3.log
: > $L
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest -m gpu ==" | tee -a $L
timeout 600 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 240 --timeout-method=thread > gpurun_out/_t.out 2>&1
grep -i -A12 "Traceback\|^E " gpurun_out/_t.out | head -40 | cut -c1-300 | tee -a $L
tail -4 gpurun_out/_t.out | tee -a $L
cp gpurun_out/gpu_ops_report.txt gpurun_out/r04_gpu_parity_report.txt 2>/dev/null
echo "== smoke ==" | tee -a $L
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $L
echo "== bench (default) ==" | tee -a $L
timeout 500 python bench.py --steps 20 --warmup 5 2> gpurun_out/r04_bench.err | tail -1 > gpurun_out/r04_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench.json'))
print('value', d['value'], d['samples'], 'min/median', d['min'] / d['median'])
print('dropin', d['dropin_surface']['value'], d['dropin_surface']['samples'])
print('roofline', {k: d['roofline'][k] for k in ('achieved','frac','mean_launch_ms')}, 'match', d['roofline_match_top2']['frac'])
print('whole_path', d['whole_path']['frac_of_f16_mfma_peak'], 'first_frame_ms', d['first_frame_ms'])
print('other', {k: (v.get('value'), v.get('samples')) for k, v in d.get('other_configs', {}).items()})
print('cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('seconds_per_frame'))
w=d.get('wavefront_model', {}); print('phases', w.get('phase_ms_per_frame_measured'))
e=w.get('predicted_speedup', {}).get('8', {}); print('wf8', {k: (v.get('speedup'), (v.get('with_context_exchange') or {}).get('speedup')) for k, v in e.items()})
" 2>&1 | cut -c1-900 | tee -a $L
