#!/usr/bin/env python3
"""The specialised 24 -> 24 3x3 conv (refvsr_conv24) per map size: with a residual (ResList tail / feat_fusion*.1 form) and
plain, alpha-gated), time per launch and GB/s of map bytes moved."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from refvsr_amd import ops  # noqa: E402
from refvsr_amd.packing import pack_conv  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
C = 24
wt = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
cw = ops.ConvWeights(pack_conv(wt, torch.zeros(C), [C]), dev)
assert cw.blob24 is not None
for name, h, w in (('LR 270x480', 270, 480), ('2x 540x960', 540, 960), ('HR 1080x1920', 1080, 1920)):
    xs = [ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev)) for _ in range(3)]
    for label, kw in (('plain', {}), ('+ residual', {'res': xs[2]}), ('* alpha + residual', {'mul': xs[1], 'res': xs[2]})):
        def run():
            y = xs[0]
            for _ in range(20):
                y = ops.conv(cw, y, act=0.2, **kw)
            return y
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ts = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        nb = 2.0 * h * w * C * 2 * (1 + len(kw) * 0.5)
        print('conv24 %-13s %-20s %7.2f us/launch  %6.0f GB/s of map bytes' % (name, label, min(ts), nb / min(ts) / 1e3), flush=True)
