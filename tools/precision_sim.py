#!/usr/bin/env python3
"""CPU model of the HIP engine's rounding points (NOT product code, NOT the oracle): the oracle's
algorithm with fp16 rounding inserted where the kernels round, to decide which tensors need fp32
storage to meet the 1e-3 dB PSNR parity bar.  Usage: python tools/precision_sim.py [frames] [h] [w]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import refvsr_oracle as orc
from refvsr_amd import get_config, make_state_dict
from refvsr_amd.synth import make_clip, window_indices

q = lambda x: x.half().float()


class Sim(object):
    def __init__(self, cfg, sd, trunk_fp32, out_fp32=False, spynet16=True):
        self.cfg, self.W = cfg, {k: v.float() for k, v in sd.items()}
        self.Wq = {k: (q(v) if k.endswith('weight') else v) for k, v in self.W.items()}
        self.qt = (lambda x: x) if trunk_fp32 else q          # trunk / state storage
        self.qo = (lambda x: x) if out_fp32 else q            # ordinary feature-map storage
        self.spynet16 = spynet16
        self.C, self.nb = cfg.mid_channels, cfg.num_blocks
        self.state = None
        self.itr = 0

    def conv(self, x, name, stride=1):       # operands rounded to fp16, fp32 accumulate, NO output rounding
        w = self.Wq['Network.' + name + '.weight']
        return F.conv2d(q(x), w, self.W['Network.' + name + '.bias'], stride=stride, padding=w.shape[-1] // 2)

    def res_list(self, x, name, n):
        x0 = x
        for i in range(n):
            t = self.qo(orc.lrelu(self.conv(x, '%s.RBs.%d.conv1' % (name, i)), 0.2))
            x = self.qt(x + self.conv(t, '%s.RBs.%d.conv2' % (name, i)))
        return self.qt(x0 + self.conv(x, name + '.conv_tail'))

    def resblocks(self, lr, feat, name):
        x = self.qt(orc.lrelu(self.conv(torch.cat([lr, feat], 1), name + '.main.0'), 0.1))
        for i in range(self.nb):
            t = self.qo(F.relu(self.conv(x, '%s.main.2.%d.conv1' % (name, i))))
            x = self.qt(x + self.conv(t, '%s.main.2.%d.conv2' % (name, i)))
        return x

    def basic2_alpha(self, pair, name):
        a = self.qo(orc.lrelu(orc.conv(pair, self.W, 'Network.' + name + '.0.0'), 0.2))     # fp32 direct conv
        return self.qo(orc.lrelu(self.conv(a, name + '.1.0'), 0.2))

    def spynet(self, ref, supp):
        if not self.spynet16:
            return orc.spynet(ref, supp, self.W)
        n, _, h, w = ref.shape
        w_up = w if w % 32 == 0 else 32 * (w // 32 + 1)
        h_up = h if h % 32 == 0 else 32 * (h // 32 + 1)
        mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1); std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        r = [(orc.resize(ref, (h_up, w_up), 'bilinear') - mean) / std]; s = [(orc.resize(supp, (h_up, w_up), 'bilinear') - mean) / std]
        for _ in range(5):
            r.append(orc.avg_pool2(r[-1])); s.append(orc.avg_pool2(s[-1]))
        r, s = r[::-1], s[::-1]
        flow = torch.zeros(n, 2, h_up // 32, w_up // 32)
        for lvl in range(6):
            fu = flow if lvl == 0 else orc.flow_up2(flow)
            x = q(torch.cat([r[lvl], orc.flow_warp_border(s[lvl], fu), fu], 1))
            for j in range(5):
                x = self.conv(x, 'FlowNet.basic_module.%d.basic_module.%d.conv' % (lvl, j))
                if j < 4:
                    x = q(F.relu(x))
            flow = fu + x
        flow = orc.resize(flow, (h, w), 'bilinear')
        return flow * torch.tensor([float(w) / w_up, float(h) / h_up]).view(1, 2, 1, 1)

    def prepare(self, lr, ref):
        conf, idx = orc.feature_match(lr, ref, self.W, False)
        h, w = lr.shape[-2:]
        x = self.qo(orc.lrelu(self.conv(ref, 'ref_encoder1.0.0'), 0.2)); x = self.qo(orc.lrelu(self.conv(x, 'ref_encoder1.1.0'), 0.2))
        rf = self.res_list(x, 'res1', 4)
        x = self.qo(orc.lrelu(self.conv(rf, 'ref_encoder2.0.0', 2), 0.2)); x = self.qo(orc.lrelu(self.conv(x, 'ref_encoder2.1.0'), 0.2))
        rfd = self.res_list(x, 'res2', 4)
        aligned = self.qo(orc.block_gather(rfd, idx, 1, (h, w)))
        feats2 = self.qo(orc.block_gather(rf, idx, 2, (2 * h, 2 * w)))
        rgb2 = q(orc.block_gather(ref, idx, 2, (2 * h, 2 * w)))
        def enc(z):
            e = self.qo(orc.lrelu(self.conv(z, 'aa2.align.conv1.0'), 0.2))
            t = self.qo(orc.lrelu(self.conv(e, 'aa2.align.conv1.2.conv1'), 0.2))
            return self.qo(orc.lrelu(e + self.conv(t, 'aa2.align.conv1.2.conv2'), 0.2))
        qy = enc(q(orc.bicubic_scale(lr, 2, False))); r = enc(rgb2)
        a = self.qo(orc.lrelu(self.conv(torch.cat([r, qy], 1), 'aa2.align.p_conv.0', 2), 0.2))
        t = self.qo(orc.lrelu(self.conv(a, 'aa2.align.p_conv.2.conv1'), 0.2))
        a = self.qo(orc.lrelu(a + self.conv(t, 'aa2.align.p_conv.2.conv2'), 0.2))
        aff = (self.conv(a, 'aa2.align.p_conv.4') + 1).clamp(-3, 3)
        aligned_up = self.qo(orc.aligned_sample(feats2, aff, 2))
        return dict(conf=conf, aligned=aligned, aligned_up=aligned_up)

    def rap(self, fr, lr, conf_prop, feat, feat_up):
        pair = torch.cat([conf_prop, fr['conf']], 1)
        alpha = self.basic2_alpha(pair, 'conf_fusion')
        t = self.qo(orc.lrelu(self.conv(torch.cat([feat, fr['aligned']], 1), 'feat_fusion.0.0'), 0.2))
        feat = self.qt(feat + alpha * orc.lrelu(self.conv(t, 'feat_fusion.1.0'), 0.2))
        feat = self.res_list(feat, 'feat_decoder', 8)
        y = self.conv(feat, 'upsample1.upsample_conv'); n, c4, h, w = y.shape
        up1 = self.qo(F.pixel_shuffle(y, 2))
        feat_up = self.qt(orc.lrelu(self.conv(torch.cat([feat_up, up1], 1), 'feat_fusion2_1.0.0'), 0.2))
        alpha2 = self.basic2_alpha(orc.bicubic_scale(pair, 2, True), 'conf_fusion2')
        t = self.qo(orc.lrelu(self.conv(torch.cat([feat_up, fr['aligned_up']], 1), 'feat_fusion2.0.0'), 0.2))
        feat_up = self.qt(feat_up + alpha2 * orc.lrelu(self.conv(t, 'feat_fusion2.1.0'), 0.2))
        feat_up = self.res_list(feat_up, 'feat_decoder2', 4)
        return feat, feat_up, torch.maximum(conf_prop, fr['conf'])

    def compute_up(self, bw, fw, cb, cf, lr):
        cat = torch.cat([bw, fw], 1)
        fus = self.qt(self.conv(cat, 'fusion_UP'))
        alpha = self.basic2_alpha(orc.bicubic_scale(torch.cat([cb, cf], 1), 2, True), 'conf_fusion_BWFW')
        t = self.qo(orc.lrelu(self.conv(cat, 'feat_fusion_BWFW.0.0'), 0.2))
        out = self.qt(fus + alpha * orc.lrelu(self.conv(t, 'feat_fusion_BWFW.1.0'), 0.2))
        out = self.res_list(out, 'feat_decoder_BWFW', 4)
        out = self.qo(F.pixel_shuffle(orc.lrelu(self.conv(out, 'upsample2.upsample_conv'), 0.1), 2))
        out = self.qo(orc.lrelu(self.conv(out, 'conv_hr'), 0.1))
        return (self.conv(out, 'conv_last') + orc.bicubic_scale(lr, 4, True)).clamp(0, 1)

    def forward(self, lrs, refs, first):
        n, t, c, h, w = lrs.shape; C = self.C; ctr = t // 2
        R = self.cfg.reset_branch
        if R is not None and self.itr == R: first = True
        fr = [self.prepare(lrs[:, i], refs[:, i]) if (first or i >= ctr) else None for i in range(t)]
        warp = lambda x, fl: self.qt(orc.warp(x, fl))
        feat = torch.zeros(n, C, h, w); feat_up = torch.zeros(n, C, 2 * h, 2 * w); conf = torch.zeros(n, 1, h, w)
        for i in range(t - 1, ctr - 1, -1):
            if i < t - 1:
                fl = self.spynet(lrs[:, i], lrs[:, i + 1])
                feat = warp(feat, fl); conf = orc.warp(conf, fl); feat_up = warp(feat_up, orc.flow_up2(fl))
            feat = self.resblocks(q(lrs[:, i]), feat, 'backward_resblocks')
            feat, feat_up, conf = self.rap(fr[i], lrs[:, i], conf, feat, feat_up)
        bw, cb = feat_up, conf
        if first:
            feat = torch.zeros(n, C, h, w); feat_up = torch.zeros(n, C, 2 * h, 2 * w); conf = torch.zeros(n, 1, h, w); rs = 0
        else:
            rs = ctr
        for i in range(rs, ctr + 1):
            if i > rs:
                fl = self.spynet(lrs[:, i], lrs[:, i - 1])
                feat = warp(feat, fl); feat_up = warp(feat, orc.flow_up2(fl)); conf = orc.warp(conf, fl)
            elif not first:
                pf, pfl, pfu, pc = self.state
                feat = warp(pf, pfl); feat_up = warp(pfu, orc.flow_up2(pfl)); conf = orc.warp(pc, pfl)
            feat = self.resblocks(q(lrs[:, i]), feat, 'forward_resblocks')
            feat, feat_up, conf = self.rap(fr[i], lrs[:, i], conf, feat, feat_up)
            if i == ctr:
                self.state = (feat, self.spynet(lrs[:, ctr + 1], lrs[:, ctr]), feat_up, conf)
        out = self.compute_up(bw, feat_up, cb, conf, lrs[:, ctr])
        if first: self.itr = 0
        self.itr += 1
        return out


def psnr(a, b):
    return 10 * math.log10(1.0 / float(((a.double() - b.double()) ** 2).mean()))


if __name__ == '__main__':
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    torch.set_num_threads(8)
    cfg = get_config('p', 'm', 'config_RefVSR_small_L1'); cfg.frame_num = 5
    sd = make_state_dict(cfg, 1234)
    lr, rf, gt = make_clip(nf, h, w, seed=5)
    o = orc.OracleNetwork(cfg, sd)
    sims = {'A all-fp16': Sim(cfg, sd, False), 'B trunk-fp32': Sim(cfg, sd, True), 'C all-fp32-store': Sim(cfg, sd, True, True),
            'D trunk32+spynet32': Sim(cfg, sd, True, False, False)}
    with torch.no_grad():
        for f in range(nf):
            wi = window_indices(f, nf, 5)
            want = o.forward(lr[wi][None], rf[wi][None], f == 0)['result']
            line = 'f%d ' % f
            for k, s in sims.items():
                got = s.forward(lr[wi][None], rf[wi][None], f == 0)
                line += '| %s: max %.2e psnr %.1f dP %.2e ' % (k, float((got - want).abs().max()), psnr(got, want),
                                                              abs(psnr(got, gt[f][None]) - psnr(want, gt[f][None])))
            print(line, flush=True)
