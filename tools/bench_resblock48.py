#!/usr/bin/env python3
"""The 48-channel residual blocks of RefVSR_MFID / RefVSR_MFID_8K: refvsr_resblock48_chain (one launch per block, weight sets
swapped per tile by LDS-DMA) against the round-3 path (two refvsr_conv48 launches per block), chains of 10 blocks queued behind
a long blocker kernel.  Per block: device microseconds, useful TFLOP/s (2 x 9 x 48 x 48 x 2 convs per pixel), share of the dense
fp16 MFMA peak, and the two paths' outputs compared element by element."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from refvsr_amd import ops  # noqa: E402
from refvsr_amd.packing import pack_conv  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    blocker = torch.randn(8192, 8192, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _ = blocker @ blocker
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    C, n = 48, 10
    pairs = []
    for _ in range(n):
        pairs.append(tuple(ops.ConvWeights(pack_conv(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5 * 0.5,
                                                     torch.randn(C, generator=g) * 0.01, [C]), dev) for _ in range(2)))
    ch = ops.Resblock48Chain(pairs, dev)

    def two_launch(x, act):
        for c1, c2 in pairs:
            x = ops.conv(c2, ops.conv(c1, x, act=act), res=x)
        return x
    for name, h, w, iters in (('LR 270x480', 270, 480, 8), ('LR/2 135x240', 135, 240, 8), ('1080x1920', 1080, 1920, 2)):
        x = ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev))
        fl = 2.0 * h * w * C * C * 9 * 2
        a, b = two_launch(x, 0.0), ops.resblock48_chain(ch, x, 0.0)
        same = bool(torch.equal(a, b))
        for label, fn in (('two conv48 launches', lambda: two_launch(x, 0.0)), ('resblock48 (fused)', lambda: ops.resblock48_chain(ch, x, 0.0)),
                          ('resblock48 lrelu', lambda: ops.resblock48_chain(ch, x, 0.2))):
            us = timeit(fn, iters) / n
            print('resblock48 %-13s %-22s %8.2f us/block  %7.1f TFLOP/s useful  %5.1f %% of 2.5 PF   fused == two launches: %s'
                  % (name, label, us, fl / us / 1e6, fl / us / 1e6 / 25.0, same), flush=True)


if __name__ == '__main__':
    main()
