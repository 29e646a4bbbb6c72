#!/usr/bin/env python3
"""Fused 24-channel residual blocks: the compile-time-specialised kernel (refvsr_resblock24_chain, 8 / 4 waves) against the
runtime-generic lean kernel (refvsr_resblock_chain), as chains of 24 blocks queued behind a long blocker kernel so that the host
is out of the picture.  Per block: device microseconds, useful TFLOP/s (2 x 9 x 24 x 24 x 2 convs per pixel) and the share
of the dense fp16 MFMA peak (2.5 PFLOP/s)."""
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
    C, n = 24, 24
    raw, pairs = [], []
    for _ in range(n):
        ws = [torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5 * 0.5 for _ in range(2)]
        bs = [torch.zeros(C) for _ in range(2)]
        raw.append(((ws[0], bs[0]), (ws[1], bs[1])))
        pairs.append(tuple(ops.ConvWeights(pack_conv(ws[i], bs[i], [C]), dev) for i in range(2)))
    ch24, chl = ops.Resblock24Chain(raw, dev), ops.ResblockChain(pairs)
    lib = ops.hip.lib()
    sizes = [('LR 270x480', 270, 480), ('2x 540x960', 540, 960), ('LR/2 135x240', 135, 240), ('HR 1080x1920', 1080, 1920)]
    iters = int(os.environ.get('RB_ITERS', '10'))
    for name, h, w in sizes:
        x = ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev))
        fl = 2.0 * h * w * C * C * 9 * 2
        res = []
        for label, fn in (('lean (generic)', lambda: ops.resblock_chain(chl, x, 0.0)),
                          ('rb24 8 waves', lambda: (lib.refvsr_set_resblock24_waves(8), ops.resblock24_chain(ch24, x, 0.0))),
                          ('rb24 4 waves', lambda: (lib.refvsr_set_resblock24_waves(4), ops.resblock24_chain(ch24, x, 0.0))),
                          ('rb24 16 waves 16x32', lambda: (lib.refvsr_set_resblock24_waves(16), ops.resblock24_chain(ch24, x, 0.0))),
                          ('rb24 default', lambda: (lib.refvsr_set_resblock24_waves(0), ops.resblock24_chain(ch24, x, 0.0))),
                          ('rb24 8 waves lrelu', lambda: (lib.refvsr_set_resblock24_waves(8), ops.resblock24_chain(ch24, x, 0.2))),
                          ('rb24 default, 16 B stores', lambda: (lib.refvsr_set_resblock24_waves(0), lib.refvsr_set_resblock24_store(1), ops.resblock24_chain(ch24, x, 0.0), lib.refvsr_set_resblock24_store(0)))):
            us = timeit(fn, iters) / n
            res.append((label, us))
            print('resblock %-14s %-26s %8.2f us/block  %7.1f TFLOP/s useful  %5.1f %% of 2.5 PF' %
                  (name, label, us, fl / us / 1e6, fl / us / 1e6 / 25.0), flush=True)
        lib.refvsr_set_resblock24_waves(0)


if __name__ == '__main__':
    main()
