#!/usr/bin/env python3
"""The strided 5x5 predictor conv of AlignedConv2d (alignment.py:20, 32 + 32 -> 32, stride 2 on the 540 x 960 encodings): device
microseconds with 32 (mt = 2: gather mode) and 16 (mt = 1: 4 x 32 tile mode) output channels per workgroup, and bit-identity of the two."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from refvsr_amd import ops  # noqa: E402
from refvsr_amd.packing import pack_conv  # noqa: E402

dev = torch.device('cuda:0')


def us(fn, iters=20):
    import gc
    gc.collect()
    gc.disable()                         # a gen-2 collection inside the timed loop stalls the host for tens of ms: the GPU
    try:                                 # idles and the events report milliseconds per launch (seen twice in round 5)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    finally:
        gc.enable()


def main():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(32, 64, 5, 5, generator=g) * 0.03
    b = torch.randn(32, generator=g) * 0.1
    for (h, wd, st) in ((540, 960, 2), (1080, 1920, 4), (61, 93, 2)):
        r = ops.pack_nhwc16(torch.randn(32, h, wd, generator=g).to(dev))
        q = ops.pack_nhwc16(torch.randn(32, h, wd, generator=g).to(dev))
        outs, ts = [], []
        for mt in (2, 1):
            cw = ops.ConvWeights(pack_conv(w, b, [32, 32], mt=mt), dev)
            outs.append(ops.conv(cw, r, q, stride=st, act=0.2))
            ts.append(us(lambda: ops.conv(cw, r, q, stride=st, act=0.2)))
        print('p_conv.0 %4dx%-4d stride %d: mt=2 %.1f us  mt=1 %.1f us  bit-identical=%s' % (h, wd, st, ts[0], ts[1], torch.equal(outs[0], outs[1])), flush=True)


if __name__ == '__main__':
    main()
