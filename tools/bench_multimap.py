#!/usr/bin/env python3
"""Multi-map launches (ABI 11): device microseconds PER MAP of the fused 24-channel block chain and of the conv24 shapes of a
propagation step with B = 1 .. 4 maps per launch, queued behind a long blocker kernel (host out of the picture).  The question
the numbers answer: what does a second / third / fourth tile per weight fill buy at 270 x 480 (one 8 x 32 tile per workgroup)?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from refvsr_amd import ops  # noqa: E402
from refvsr_amd.packing import pack_conv  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters):
    import gc
    gc.collect()
    gc.disable()                         # a gen-2 collection inside the timed loop stalls the host for tens of ms: the GPU
    try:                                 # idles and the events report milliseconds per launch (seen twice in round 5)
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
    finally:
        gc.enable()


def main():
    g = torch.Generator().manual_seed(0)
    C, n = 24, 24
    raw = []
    for _ in range(n):
        ws = [torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5 * 0.5 for _ in range(2)]
        raw.append(((ws[0], torch.zeros(C)), (ws[1], torch.zeros(C))))
    ch = ops.Resblock24Chain(raw, dev)
    iters = int(os.environ.get('RB_ITERS', '10'))
    sizes = [('LR 270x480', 270, 480), ('LR/2 135x240', 135, 240), ('2x 540x960', 540, 960)]
    for name, h, w in ([] if os.environ.get('RB48_ONLY', '0') == '1' else sizes):
        fl = 2.0 * h * w * C * C * 9 * 2
        for B in (1, 2, 3, 4):
            xs = [ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev)) for _ in range(B)]
            fn = (lambda: ops.resblock24_chain(ch, xs[0], 0.0)) if B == 1 else (lambda: ops.resblock24_chain_b(ch, xs, 0.0))
            us = timeit(fn, iters) / n
            print('multimap resblock24 %-13s B=%d  %7.2f us/launch  %6.2f us/map  %6.1f TFLOP/s useful  %5.1f %% of 2.5 PF' %
                  (name, B, us, us / B, fl * B / us / 1e6, fl * B / us / 1e6 / 25.0), flush=True)
    if os.environ.get('RB48', '1') != '0':            # the 48-channel block (ABI 12): launch cost + first fill + tail shared, sets swap per tile
        C48, n48 = 48, 12
        raw = []
        for _ in range(n48):
            ws = [torch.randn(C48, C48, 3, 3, generator=g) / (C48 * 9) ** 0.5 * 0.5 for _ in range(2)]
            raw.append(((ws[0], torch.zeros(C48)), (ws[1], torch.zeros(C48))))
        ch48 = ops.Resblock48Chain(raw, dev)
        for name, h, w in sizes:
            fl = 2.0 * h * w * C48 * C48 * 9 * 2
            for B in (1, 2, 3, 4):
                xs = [ops.pack_nhwc16(torch.randn(C48, h, w, generator=g).to(dev)) for _ in range(B)]
                fn = (lambda: ops.resblock48_chain(ch48, xs[0], 0.0)) if B == 1 else (lambda: ops.resblock48_chain_b(ch48, xs, 0.0))
                us = timeit(fn, iters) / n48
                print('multimap resblock48 %-13s B=%d  %7.2f us/launch  %6.2f us/map  %6.1f TFLOP/s useful  %5.1f %% of 2.5 PF' %
                      (name, B, us, us / B, fl * B / us / 1e6, fl * B / us / 1e6 / 25.0), flush=True)
    if os.environ.get('RB48_ONLY', '0') == '1':
        return
    gq = torch.Generator().manual_seed(1)
    shapes = [('24->24', [24], False), ('8+24->24', [8, 24], False), ('24+24->24', [24, 24], False), ('24->96 shuffle', [24], True)]
    for name, h, w in sizes[:1] + sizes[2:]:
        for sname, cins, shuf in shapes:
            co = 96 if shuf else 24
            cw = ops.ConvWeights(pack_conv(torch.randn(co, sum(cins), 3, 3, generator=gq) * 0.05, torch.zeros(co), cins, shuf), dev)
            for B in (1, 2, 4):
                s0 = [ops.pack_nhwc16(torch.randn(cins[0], h, w, generator=gq).to(dev)) for _ in range(B)]
                s1 = [ops.pack_nhwc16(torch.randn(cins[1], h, w, generator=gq).to(dev)) for _ in range(B)] if len(cins) > 1 else None
                if B == 1:
                    fn = lambda: [ops.conv(cw, s0[0], None if s1 is None else s1[0], act=0.2) for _ in range(8)]
                else:
                    fn = lambda: [ops.conv_b(cw, s0, s1, act=0.2) for _ in range(8)]
                us = timeit(fn, iters) / 8
                print('multimap conv %-14s %-13s B=%d  %7.2f us/launch  %6.2f us/map' % (sname, name, B, us, us / B), flush=True)


if __name__ == '__main__':
    main()
