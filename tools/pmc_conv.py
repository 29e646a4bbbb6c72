#!/usr/bin/env python3
"""A few launches of the throughput-bound conv shapes for rocprofv3 --pmc (LDS / MFMA counters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refvsr_amd import ops
from refvsr_amd.packing import pack_conv
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
C = 24
w1 = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
c1 = ops.ConvWeights(pack_conv(w1, torch.zeros(C), [C]), dev)
c2 = ops.ConvWeights(pack_conv(w1.flip(0), torch.zeros(C), [C]), dev)
x = ops.pack_nhwc16(torch.randn(C, 540, 960, generator=g).to(dev))
for _ in range(3):
    ops.conv(c1, x, act=0.2)
for _ in range(3):
    ops.resblock(c1, c2, x, act=0.0)
torch.cuda.synchronize()
print('done')
