#!/usr/bin/env python3
"""Runs the dominant kernel (match_top2) a few times at the BASELINE size so that rocprofv3 --pmc can
attribute HBM traffic counters to it (one counter set per run, see tools/gpu_runs/final.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refvsr_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
h, w = 270, 480
lr_f = torch.randn(16, h, w, generator=g).to(dev)
ref_f = torch.randn(16, h // 2, w // 2, generator=g).to(dev)
lr_rows, _ = ops.match_patches(lr_f, 512)
ref_rows, _ = ops.match_patches(ref_f, 256)
for _ in range(4):
    ops.match_top2(ref_rows, (h // 2) * (w // 2), lr_rows, h * w, 1)
torch.cuda.synchronize()
print('done')
