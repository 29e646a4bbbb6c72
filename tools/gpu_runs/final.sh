#!/bin/bash
# Round-end style verification: full GPU test suite, smoke, bench (with cpu baseline), rocprofv3 stats, PMC traffic.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/gpu_ops_report.txt
echo "== pytest gpu ==" | tee gpurun_out/final.log
timeout 1200 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider -n 4 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "^(FAILED|ERROR|SKIPPED)|passed|failed" gpurun_out/pytest_gpu_full.log | tail -30 | tee -a gpurun_out/final.log
grep -E "^E  " gpurun_out/pytest_gpu_full.log | head -40 | tee -a gpurun_out/final.log
echo "== smoke ==" | tee -a gpurun_out/final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a gpurun_out/final.log
echo "== pmc (match_top2) ==" | tee -a gpurun_out/final.log
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_fetch" -o m -- python "$OLDPWD/tools/pmc_match.py" > "$OLDPWD/gpurun_out/pmc_fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_write" -o m -- python "$OLDPWD/tools/pmc_match.py" > "$OLDPWD/gpurun_out/pmc_write.log" 2>&1)
python tools/pmc_to_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_match_top2.json | tee -a gpurun_out/final.log
cp gpurun_out/pmc_match_top2.json profiles/pmc_match_top2.json
echo "== bench ==" | tee -a gpurun_out/final.log
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.log
echo "== bench --no-cache ==" | tee -a gpurun_out/final.log
timeout 300 python bench.py --no-cache --no-cpu-baseline 2>&1 | tail -1 | cut -c1-240 | tee gpurun_out/bench_nocache.log
echo "== rocprof ==" | tee -a gpurun_out/final.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200
for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -12 "$f" | cut -c1-150; done
