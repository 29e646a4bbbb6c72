#!/bin/bash
# where do match_top2's wave cycles go? (SQ wait / issue-stall / active split, MFMA busy, LDS conflicts)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_match_sq
(cd /tmp && timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OLDPWD/gpurun_out/pmc_match_sq" -o m -- python "$OLDPWD/tools/pmc_match.py" > "$OLDPWD/gpurun_out/pmc_match_sq.log" 2>&1)
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/pmc_match_sq/m_counter_collection.csv')))
acc = collections.defaultdict(list)
for r in rows:
    if 'match_top2' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
wc = m['SQ_WAVE_CYCLES']
print('match_top2: parked %.0f%%  issue-stall %.0f%%  issuing %.0f%%;  MFMA busy cycles %.3g;  LDS conflict %.1f%% of LDS cycles'
      % (100 * m['SQ_WAIT_ANY'] / wc, 100 * m['SQ_WAIT_INST_ANY'] / wc, 100 * m['SQ_ACTIVE_INST_ANY'] / wc,
         m['SQ_VALU_MFMA_BUSY_CYCLES'], 100 * m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1)))
PY
