#!/usr/bin/env python3
"""Driver for the rocprofv3 --pmc passes of the kernels bench.py lists under `kernels` (HBM-bound warp / gather /
sampler / resize kernels and the time-dominant conv kernels) plus match_top2, each launched a few times at the BASELINE
size (270x480, RefVSR_small).  A marker launch of refvsr_max2 on an n-element buffer precedes each group so that the
post-processing (tools/pmc_to_json.py) can attribute dispatches by order.  One counter set per run:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_fetch -o k -- python tools/pmc_kernels.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out_write -o k -- python tools/pmc_kernels.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from refvsr_amd import get_config, make_state_dict, ops  # noqa: E402
from refvsr_amd.engine import Engine, Weights  # noqa: E402

dev = torch.device('cuda:0')
GROUPS = ['resblock LR', 'resblock 2x', 'conv HR', 'conv shuffle 2x', 'warp LR', 'warp 2x', 'gather 2x', 'aligned_sample 2x',
          'bicubic x4', 'match_top2', 'warp up2 2x', 'conf_alpha LR', 'conf_alpha 2x', 'resblock LR x4 maps', 'resblock LR x2 maps']
REPS = 4


def main():
    cfg = get_config('p', 'm', 'config_RefVSR_small_L1')
    cfg.frame_num = 5
    eng = Engine(cfg, Weights(cfg, make_state_dict(cfg, 1234), dev))
    h, w, C = 270, 480, cfg.mid_channels
    g = torch.Generator().manual_seed(3)
    rnd16 = lambda hh, ww, c: ops.pack_nhwc16(torch.randn(c, hh, ww, generator=g).to(dev))
    x_lr, x_2x, x_hr = rnd16(h, w, C), rnd16(2 * h, 2 * w, C), rnd16(4 * h, 4 * w, C)
    xs4 = [rnd16(h, w, C) for _ in range(4)]
    flow = (torch.randn(2, h, w, generator=g) * 2).to(dev)
    flow2 = ops.flow_up2(flow)
    idx = torch.randint(0, (h // 2) * (w // 2), (h * w,), generator=g, dtype=torch.int32).to(dev)
    aff = (torch.rand(3, h, w, generator=g) * 0.4 + 0.8).to(dev)
    lr = torch.rand(3, h, w, generator=g).to(dev)
    lr_f = torch.randn(16, h, w, generator=g).to(dev)
    ref_f = torch.randn(16, h // 2, w // 2, generator=g).to(dev)
    lr_rows, _ = ops.match_patches(lr_f, 512)
    ref_rows, _ = ops.match_patches(ref_f, 256)
    ca, cb = torch.rand(1, h, w, generator=g).to(dev), torch.rand(1, h, w, generator=g).to(dev)
    w0, b0 = eng.W.raw['conf_fusion2.0.0']
    cwa = eng.cw('conf_fusion2.1.0')
    nb = cfg.num_blocks
    c1, c2 = eng.cw('backward_resblocks.main.2.%d.conv1' % (nb // 2)), eng.cw('backward_resblocks.main.2.%d.conv2' % (nb // 2))
    d1, d2 = eng.cw('feat_decoder2.RBs.1.conv1'), eng.cw('feat_decoder2.RBs.1.conv2')
    fns = {
        'resblock LR': lambda: eng._block_chain(x_lr, [(c1, c2)], 0.0),         # the engine's dispatch: resblock24 for C = 24
        'resblock 2x': lambda: eng._block_chain(x_2x, [(d1, d2)], 0.2),
        'conv HR': lambda: ops.conv(eng.cw('conv_hr'), x_hr, act=0.1),
        'conv shuffle 2x': lambda: ops.conv(eng.cw('upsample2.upsample_conv'), x_2x, act=0.1),
        'warp LR': lambda: ops.warp_nhwc16(x_lr, flow),
        'warp 2x': lambda: ops.warp_nhwc16(x_2x, flow2),
        'gather 2x': lambda: ops.block_gather_nhwc16(x_lr, idx, h, w, 2),
        'aligned_sample 2x': lambda: ops.aligned_sample(x_2x, aff, 2),
        'bicubic x4': lambda: ops.bicubic_scale(lr, 4, clamp01=True),
        'match_top2': lambda: ops.match_top2(ref_rows, (h // 2) * (w // 2), lr_rows, h * w, 1),
        'warp up2 2x': lambda: ops.warp_nhwc16_up2(x_2x, flow),
        'conf_alpha LR': lambda: ops.conf_alpha(ca, cb, 1, w0, b0, cwa, want_max=True),
        'conf_alpha 2x': lambda: ops.conf_alpha(ca, cb, 2, w0, b0, cwa),
        # round 5: the multi-map launches of a frame group (four / two maps behind one launch, refvsr_resblock24_chain_batch)
        'resblock LR x4 maps': lambda: eng._block_chain_b(xs4, [(c1, c2)], 0.0),
        'resblock LR x2 maps': lambda: eng._block_chain_b(xs4[:2], [(c1, c2)], 0.0),
    }
    torch.cuda.synchronize()
    for gi, name in enumerate(GROUPS):
        m = torch.zeros(64 * (gi + 1), device=dev)
        ops.max2(m, m)                               # marker: max2 launch whose size encodes the group
        torch.cuda.synchronize()
        for _ in range(REPS):
            fns[name]()
        torch.cuda.synchronize()
    print('done')


if __name__ == '__main__':
    main()
