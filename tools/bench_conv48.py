#!/usr/bin/env python3
"""C = 48 conv shapes of RefVSR_MFID / RefVSR_MFID_8K (the one-workgroup-per-CU cases of conv_mfma.hip), timed behind a
long blocker kernel so that the host is out of the picture.  A/B knobs: REFVSR_CONV_NO_NW8, REFVSR_CONV_NO_W16."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from refvsr_amd import ops  # noqa: E402
from refvsr_amd.packing import pack_conv  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=30):
    for _ in range(3):
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
    g = torch.Generator().manual_seed(1)
    f32_cases = [('vgg f32 3->64', 64, [3], 270, 480), ('vgg f32 64->64', 64, [64], 270, 480), ('vgg f32 64->64 ref', 64, [64], 135, 240)]
    cases = [('LR270 48->48', 48, [48], 270, 480), ('LR270 48+48->48', 48, [48, 48], 270, 480),
             ('2x540 48->48', 48, [48], 540, 960), ('2x540 48+48->48', 48, [48, 48], 540, 960),
             ('LR1080 48->48', 48, [48], 1080, 1920), ('2x2160 48+48->48', 48, [48, 48], 2160, 3840),
             ('LR270 64->64', 64, [64], 270, 480),
             ('S LR270 24->24', 24, [24], 270, 480), ('S LR270 24+24->24', 24, [24, 24], 270, 480),
             ('S 2x540 24->24', 24, [24], 540, 960), ('S 2x540 24+24->24', 24, [24, 24], 540, 960),
             ('S HR1080 24->24', 24, [24], 1080, 1920), ('S LR135 24->24', 24, [24], 135, 240)]
    only = os.environ.get('CONV48_ONLY')
    for name, co, cins, h, w in f32_cases:
        if only and only not in name:
            continue
        cin = sum(cins)
        wt = torch.randn(co, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        cw = ops.ConvWeights(pack_conv(wt, torch.zeros(co), cins, False, f32=True), dev)
        x = ops.pack_nhwc32(torch.randn(cin, h, w, generator=g).to(dev))
        us = timeit(lambda: ops.conv(cw, x, act=0.0), iters=int(os.environ.get('CONV48_ITERS', '30')))
        print('conv %-20s %9.1f us  %6.1f TFLOP/s (fp32 MFMA peak 157)' % (name, us, 2.0 * h * w * co * cin * 9 / us / 1e6), flush=True)
    for name, co, cins, h, w in cases:
        if only and only not in name:
            continue
        cin = sum(cins)
        wt = torch.randn(co, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        cw = ops.ConvWeights(pack_conv(wt, torch.zeros(co), cins, False), dev)
        srcs = [ops.pack_nhwc16(torch.randn(c, h, w, generator=g).to(dev)) for c in cins]
        fn = lambda: ops.conv(cw, srcs[0], srcs[1] if len(srcs) > 1 else None, act=0.2)
        us = timeit(fn, iters=int(os.environ.get('CONV48_ITERS', '30')))
        fl = 2.0 * h * w * co * cin * 9
        print('conv %-20s %9.1f us  %6.1f TFLOP/s useful (%4.1f %% of 2.5 PF; x2.67 issued with hi+lo weights and 48->48 rows)'
              % (name, us, fl / us / 1e6, fl / us / 1e6 / 25.0), flush=True)


if __name__ == '__main__':
    main()
