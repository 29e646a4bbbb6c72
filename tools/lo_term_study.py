#!/usr/bin/env python3
"""Per-layer-group sensitivity of the end-to-end PSNR to dropping the lo term of the fp16 hi+lo conv weights.
CPU model (tools/precision_sim.py: the oracle's algorithm with fp16 rounding where the kernels round); for every group
the weights of THAT group are rounded to plain fp16 and every other conv keeps exact (hi+lo-grade) weights.  Reported:
|PSNR(model, GT) - PSNR(oracle, GT)| per frame (the north-star bar is 1e-3 dB; the budget a group may use is what is
left after the activation-storage error of the 'none' row).
Usage: python tools/lo_term_study.py [frames] [h] [w] [variant]"""
import math
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F

from oracle import refvsr_oracle as orc
from refvsr_amd import get_config, make_state_dict
from refvsr_amd.synth import make_clip, window_indices
import precision_sim as ps

GROUPS = [
    ('none', r'^$'),
    ('spynet', r'^Network\.FlowNet\.'),
    ('prop_resblocks', r'^Network\.(backward|forward)_resblocks\.'),
    ('encoders_lr', r'^Network\.(ref_encoder\d|res\d)\.'),
    ('fusion_2x', r'^Network\.(conf_fusion|feat_fusion|feat_decoder)'),
    ('align', r'^Network\.aa2\.'),
    ('upsampler', r'^Network\.(fusion_UP|upsample\d|conv_hr|conv_last)'),
    ('all', r'^Network\.(?!feature_match)'),
]


class SimG(ps.Sim):
    def __init__(self, cfg, sd, pattern):
        ps.Sim.__init__(self, cfg, sd, False)
        rx = re.compile(pattern)
        self.Wq = {k: (ps.q(v) if (k.endswith('weight') and rx.search(k)) else v) for k, v in self.W.items()}

    def conv(self, x, name, stride=1):
        w = self.Wq['Network.' + name + '.weight']
        return F.conv2d(ps.q(x), w, self.W['Network.' + name + '.bias'], stride=stride, padding=w.shape[-1] // 2)


def psnr(a, b):
    return 10 * math.log10(1.0 / float(((a.double() - b.double()) ** 2).mean()))


if __name__ == '__main__':
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    variant = sys.argv[4] if len(sys.argv) > 4 else None
    if variant == 'random':
        variant = None
    torch.set_num_threads(8)
    cfg = get_config('p', 'm', 'config_RefVSR_small_L1')
    cfg.frame_num = 5
    sd = make_state_dict(cfg, 1234, variant=variant)
    lr, rf, gt = make_clip(nf, h, w, seed=5)
    o = orc.OracleNetwork(cfg, sd)
    sims = [(g, SimG(cfg, sd, p)) for g, p in GROUPS]
    worst = {g: 0.0 for g, _ in GROUPS}
    with torch.no_grad():
        for f in range(nf):
            wi = window_indices(f, nf, 5)
            want = o.forward(lr[wi][None], rf[wi][None], f == 0)['result']
            base = psnr(want, gt[f][None])
            line = 'f%d' % f
            for g, s in sims:
                got = s.forward(lr[wi][None], rf[wi][None], f == 0)
                dp = abs(psnr(got, gt[f][None]) - base)
                worst[g] = max(worst[g], dp)
                line += ' | %s %.2e' % (g, dp)
            print(line, flush=True)
    print('worst |dPSNR| per group: ' + ', '.join('%s %.2e' % (g, worst[g]) for g, _ in GROUPS))
