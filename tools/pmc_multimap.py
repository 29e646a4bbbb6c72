#!/usr/bin/env python3
"""rocprofv3 --pmc driver for the launches of record of round 5: the fused 24-channel block as a four-map LR launch (sixteen waves), as a
single-map LR launch (eight waves) and on a 2x map, the 24 + 24 -> 24 conv of the 2x fusion sites as a four-map launch, and the 48-channel
block (single map / four maps).  Six launches each; summarise with tools/pmc_summary.py (SQ counters: matrix-pipe busy cycles, LDS-array
cycles, bank conflicts, wait states)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from refvsr_amd import ops  # noqa: E402
from refvsr_amd.packing import pack_conv  # noqa: E402

dev = torch.device('cuda:0')


def main():
    g = torch.Generator().manual_seed(0)
    mk = lambda c: ((torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5 * 0.5, torch.zeros(c)),
                    (torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5 * 0.5, torch.zeros(c)))
    ch24 = ops.Resblock24Chain([mk(24)], dev)
    ch48 = ops.Resblock48Chain([mk(48)], dev)
    cw = ops.ConvWeights(pack_conv(torch.randn(24, 48, 3, 3, generator=g) * 0.05, torch.zeros(24), [24, 24], False), dev)
    m = lambda c, h, w: ops.pack_nhwc16(torch.randn(c, h, w, generator=g).to(dev))
    x4 = [m(24, 270, 480) for _ in range(4)]
    x2 = m(24, 540, 960)
    y4 = [m(48, 270, 480) for _ in range(4)]
    a4 = [m(24, 540, 960) for _ in range(4)]
    b4 = [m(24, 540, 960) for _ in range(4)]
    for fn in (lambda: ops.resblock24_chain_b(ch24, x4, 0.0),         # grid 256 x 1024 threads: the launches of record
               lambda: ops.resblock24_chain(ch24, x4[0], 0.0),        # grid 510 x 512
               lambda: ops.resblock24_chain(ch24, x2, 0.0),           # grid 256 x 1024, 2x map
               lambda: ops.conv_b(cw, a4, b4, act=0.2),
               lambda: ops.resblock48_chain(ch48, y4[0], 0.0),
               lambda: ops.resblock48_chain_b(ch48, y4, 0.0)):
        for _ in range(6):
            fn()
        torch.cuda.synchronize()
    print('done')


if __name__ == '__main__':
    main()
