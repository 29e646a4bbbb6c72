#!/usr/bin/env python3
"""match_top2 at the BASELINE size (270x480 VGG features): time per launch, % of the dense fp16 MFMA peak and checksums of
the outputs (checksums: to compare schedule variants across processes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from refvsr_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
h, w = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '270x480').split('x'))
lr_f = torch.randn(16, h, w, generator=g).to(dev)
ref_f = torch.randn(16, h, w, generator=g).to(dev)
lr_rows, inv_lr = ops.match_patches(lr_f, 512)
ref_rows, inv_ref = ops.match_patches(ref_f, 256)
n = h * w
idx, val = ops.match_top2(ref_rows, n, lr_rows, n, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record()
    for _ in range(5):
        ops.match_top2(ref_rows, n, lr_rows, n, 1)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5 * 1e3)
us = min(ts)
flops = 2.0 * n * n * 144
print('match_top2 %s (%s): %.1f us (min of 5x5; all: %s)  %.1f TFLOP/s = %.1f %% of 2.5 PF  | idx checksum %d  val checksum %.6f'
      % (sys.argv[1] if len(sys.argv) > 1 else '270x480', 'match_top2_kernel_v4', us,
         ' '.join('%.0f' % t for t in ts), flops / us / 1e6, flops / us / 1e6 / 25.0,
         int(idx.long().sum()), float(val.double().sum())))
