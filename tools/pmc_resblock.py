#!/usr/bin/env python3
"""rocprofv3 --pmc driver for the fused 24-channel ResBlock kernels: the specialised kernel (resblock24, 8 and 4 waves) and the
generic lean kernel at LR (270x480) and 2x (540x960), six launches each.  Summarise with tools/pmc_summary.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from refvsr_amd import ops  # noqa: E402
from refvsr_amd.packing import pack_conv  # noqa: E402

dev = torch.device('cuda:0')


def main():
    g = torch.Generator().manual_seed(0)
    C = 24
    ws = [torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5 * 0.5 for _ in range(2)]
    bs = [torch.zeros(C), torch.zeros(C)]
    ch24 = ops.Resblock24Chain([((ws[0], bs[0]), (ws[1], bs[1]))], dev)
    pair = tuple(ops.ConvWeights(pack_conv(ws[i], bs[i], [C]), dev) for i in range(2))
    lib = ops.hip.lib()
    for h, w in ((270, 480), (540, 960)):
        x = ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev))
        for waves in (8, 4):
            lib.refvsr_set_resblock24_waves(waves)
            for _ in range(6):
                ops.resblock24_chain(ch24, x, 0.0)
            torch.cuda.synchronize()
        lib.refvsr_set_resblock24_waves(8)
        for _ in range(6):
            ops.resblock(pair[0], pair[1], x, act=0.0)
        torch.cuda.synchronize()
    print('done')


if __name__ == '__main__':
    main()
