#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmcc1 gpurun_out/pmcc2 gpurun_out/pmcc3
(cd /tmp && timeout 100 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OLDPWD/gpurun_out/pmcc1" -o c -- python "$OLDPWD/tools/pmc_conv.py" > "$OLDPWD/gpurun_out/pmcc1.log" 2>&1)
(cd /tmp && timeout 100 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OLDPWD/gpurun_out/pmcc2" -o c -- python "$OLDPWD/tools/pmc_conv.py" > "$OLDPWD/gpurun_out/pmcc2.log" 2>&1)
(cd /tmp && timeout 100 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmcc3" -o c -- python "$OLDPWD/tools/pmc_conv.py" > "$OLDPWD/gpurun_out/pmcc3.log" 2>&1)
ls gpurun_out/pmcc1 gpurun_out/pmcc2 gpurun_out/pmcc3; tail -2 gpurun_out/pmcc1.log
