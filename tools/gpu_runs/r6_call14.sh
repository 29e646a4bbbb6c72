#!/bin/bash
# round 6, call 14: SQ counters of the multi-map launches of record (matrix-pipe busy, LDS-array cycles, wait states)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r06_pmc_sq_multimap.txt
: > $L
for pass in 1 2 3; do
  case $pass in
    1) CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES";;
    2) CTRS="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE";;
    3) CTRS="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE";;
  esac
  rm -rf gpurun_out/pmc_mm$pass
  (cd /tmp && timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d "$OLDPWD/gpurun_out/pmc_mm$pass" -o k -- python "$OLDPWD/tools/pmc_multimap.py" > "$OLDPWD/gpurun_out/pmc_mm$pass.log" 2>&1)
  echo "== pmc pass $pass: $CTRS" | tee -a $L
  python tools/pmc_summary.py gpurun_out/pmc_mm$pass "kernel" | tee -a $L
  tail -2 gpurun_out/pmc_mm$pass.log
  rm -rf gpurun_out/pmc_mm$pass
done
