#!/bin/bash
# round 5, call 20: 600-frame soak of the three call modes (bit-identity on every frame, allocator high-water mark), 270 x 480 and 64 x 96
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r05_soak.log
: > $L
timeout 900 python tools/soak.py --frames 600 2>&1 | grep -v amdgpu.ids | tee -a $L
timeout 600 python tools/soak.py --frames 600 --size 64x96 --config config_RefVSR_small_MFID 2>&1 | grep -v amdgpu.ids | tee -a $L
