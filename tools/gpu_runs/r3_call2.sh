# round 3, call 2: where does resblock24 spend its time?  s_memtime probe + SQ counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== probe"; timeout 300 python tools/probe_resblock24.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_probe_resblock24.txt
for pass in 1 2 3; do
  case $pass in
    1) CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES";;
    2) CTRS="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE";;
    3) CTRS="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL";;
  esac
  rm -rf gpurun_out/pmc_rb$pass
  (cd /tmp && timeout 200 rocprofv3 --pmc $CTRS --output-format csv -d "$OLDPWD/gpurun_out/pmc_rb$pass" -o k -- python "$OLDPWD/tools/pmc_resblock.py" > "$OLDPWD/gpurun_out/pmc_rb$pass.log" 2>&1)
  echo "== pmc pass $pass"; python tools/pmc_summary.py gpurun_out/pmc_rb$pass resblock | tee gpurun_out/r3_pmc_resblock_pass$pass.txt
  tail -2 gpurun_out/pmc_rb$pass.log
  rm -rf gpurun_out/pmc_rb$pass
done
