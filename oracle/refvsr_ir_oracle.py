"""CPU ORACLE for the RefVSR_IR inference path (IconVSR-style variant: information refill from an EDVR-M feature
extractor with PCD alignment on modulated deformable convolutions) -- TEST INFRASTRUCTURE ONLY, like refvsr_oracle.py.

Independent fp32 restatement of /root/reference/models/archs/RefVSR_IR.py (Network.forward :219-391, EDVRFeatureExtractor
:424-546) and models/archs/edvr_net.py (ModulatedDCNPack :15-56, PCDAlignment :59-185, TSAFusion :188-300).  Everything
RefVSR_IR shares with RefVSR (SPyNet, matching, alignment, RAP, upsampler) is taken from refvsr_oracle.py.

Pinning: fixtures produced by the reference itself (tools/gen_golden.py, tests/golden/e2e_IR_*.npz).  The reference's
modulated deformable convolution is mmcv-full's compiled op, absent from this image; the fixtures are generated with the
pure-PyTorch restatement of that op in tools/ref_shims/mmcv/ops (published semantics: dmcn_im2col_bilinear) -- so the
DCN itself is pinned against a restatement of mmcv, NOT against mmcv's binary: parity unpinned at that boundary.
"""
import collections

import numpy as np
import torch
import torch.nn.functional as F

from . import refvsr_oracle as orc
from .refvsr_oracle import conv, lrelu

M = 64            # EDVR-M channels
DG = 8            # deformable groups


def up2_bilinear(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) (edvr_net.py:131,246)."""
    return orc.resize(x, (2 * x.shape[-2], 2 * x.shape[-1]), 'bilinear', 0.5)


def pool3s2(x, kind):
    """MaxPool2d / AvgPool2d (3, stride 2, padding 1; the average counts the padding: count_include_pad) (edvr_net.py:214-215)."""
    n, c, h, w = x.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    fill = float('-inf') if kind == 'max' else 0.0
    xp = torch.full((n, c, h + 2, w + 2), fill, dtype=x.dtype)
    xp[:, :, 1:-1, 1:-1] = x
    taps = [xp[:, :, dy:dy + 2 * ho:2, dx:dx + 2 * wo:2][:, :, :ho, :wo] for dy in range(3) for dx in range(3)]
    st = torch.stack(taps, 0)
    return st.max(0)[0] if kind == 'max' else st.sum(0) / 9.0


def deform_sample(x, py, px):
    """Bilinear sample of x [n,c,H,W] at float coordinates py, px [n,1,H,W]; 0 outside (-1, H) x (-1, W), missing corner
    pixels contribute 0 (mmcv modulated_deform_conv_cuda_kernel.cuh: dmcn_im2col_bilinear)."""
    n, c, H, W = x.shape
    inside = (py > -1) & (px > -1) & (py < H) & (px < W)
    y0, x0 = torch.floor(py), torch.floor(px)
    ly, lx = py - y0, px - x0
    y0, x0 = y0.long(), x0.long()
    flat = x.reshape(n, c, H * W)
    out = torch.zeros_like(x)
    for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx), (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
        yy, xx = y0 + dy, x0 + dx
        ok = inside & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(n, 1, H * W).expand(n, c, H * W)
        out = out + torch.gather(flat, 2, idx).view(n, c, H, W) * (wgt * ok.to(x.dtype))
    return out


def dcn_pack(x, extra, W, name):
    """ModulatedDCNPack.forward (edvr_net.py:49-56): offsets and masks from `extra`; 3x3, stride 1, padding 1, 8 deformable
    groups.  conv_offset's 216 channels = [o1 | o2 | mask]; cat(o1, o2) is the offset tensor whose channel g*18 + 2k (+1) is
    the row (column) offset of tap k in group g; mask channel g*9 + k, through a sigmoid."""
    out = conv(extra, W, name + '.conv_offset')
    offset, mask = out[:, :2 * DG * 9], torch.sigmoid(out[:, 2 * DG * 9:])
    n, c, H, Wd = x.shape
    cg = c // DG
    ys = torch.arange(H, dtype=x.dtype).view(1, 1, H, 1)
    xs = torch.arange(Wd, dtype=x.dtype).view(1, 1, 1, Wd)
    w = W[name + '.weight']                                          # [cout, c, 3, 3]
    acc = torch.zeros(n, w.shape[0], H, Wd, dtype=x.dtype)
    for k in range(9):
        ky, kx = divmod(k, 3)
        col = torch.empty_like(x)
        for g in range(DG):
            py = ys + (ky - 1) + offset[:, g * 18 + 2 * k:g * 18 + 2 * k + 1]
            px = xs + (kx - 1) + offset[:, g * 18 + 2 * k + 1:g * 18 + 2 * k + 2]
            col[:, g * cg:(g + 1) * cg] = deform_sample(x[:, g * cg:(g + 1) * cg], py, px) * mask[:, g * 9 + k:g * 9 + k + 1]
        acc = acc + torch.einsum('nchw,oc->nohw', col, w[:, :, ky, kx])
    return acc + W[name + '.bias'].view(1, -1, 1, 1)


def pcd_alignment(nbr, ref, W, P):
    """PCDAlignment.forward (edvr_net.py:134-185); nbr / ref: [L1, L2, L3] feature lists."""
    cm = lambda x, name, act=0.1: (lrelu(conv(x, W, P + name + '.conv'), act) if act is not None else conv(x, W, P + name + '.conv'))
    up_off = up_feat = None
    feat = None
    for i in (3, 2, 1):
        lv = 'l%d' % i
        off = cm(torch.cat([nbr[i - 1], ref[i - 1]], 1), 'offset_conv1.' + lv)
        if i == 3:
            off = cm(off, 'offset_conv2.' + lv)
        else:
            off = cm(torch.cat([off, up_off], 1), 'offset_conv2.' + lv)
            off = cm(off, 'offset_conv3.' + lv)
        feat = dcn_pack(nbr[i - 1], off, W, P + 'dcn_pack.' + lv)
        if i == 3:
            feat = lrelu(feat, 0.1)
        else:
            feat = cm(torch.cat([feat, up_feat], 1), 'feat_conv.' + lv, 0.1 if i == 2 else None)
        if i > 1:
            up_off = up2_bilinear(off) * 2
            up_feat = up2_bilinear(feat)
    off = cm(cm(torch.cat([feat, ref[0]], 1), 'cas_offset_conv1'), 'cas_offset_conv2')
    return lrelu(dcn_pack(feat, off, W, P + 'cas_dcnpack'), 0.1)


def tsa_fusion(aligned, W, P, center):
    """TSAFusion.forward (edvr_net.py:248-300); aligned [n, t, c, h, w]."""
    n, t, c, h, w = aligned.shape
    cm = lambda x, name: lrelu(conv(x, W, P + name + '.conv'), 0.1)
    emb_ref = conv(aligned[:, center], W, P + 'temporal_attn1')
    emb = conv(aligned.reshape(-1, c, h, w), W, P + 'temporal_attn2').view(n, t, c, h, w)
    corr = torch.sigmoid((emb * emb_ref[:, None]).sum(2))              # [n, t, h, w]
    al = (aligned * corr[:, :, None]).reshape(n, t * c, h, w)
    feat = cm(al, 'feat_fusion')
    attn = cm(al, 'spatial_attn1')
    attn = cm(torch.cat([pool3s2(attn, 'max'), pool3s2(attn, 'avg')], 1), 'spatial_attn2')
    lvl = cm(attn, 'spatial_attn_l1')
    lvl = cm(torch.cat([pool3s2(lvl, 'max'), pool3s2(lvl, 'avg')], 1), 'spatial_attn_l2')
    lvl = up2_bilinear(cm(lvl, 'spatial_attn_l3'))
    attn = cm(attn, 'spatial_attn3') + lvl
    attn = up2_bilinear(cm(attn, 'spatial_attn4'))
    attn = conv(attn, W, P + 'spatial_attn5')
    attn_add = conv(cm(attn, 'spatial_attn_add1'), W, P + 'spatial_attn_add2')
    return feat * torch.sigmoid(attn) * 2 + attn_add


def edvr_features(x, W, P='Network.edvr.', center=2):
    """EDVRFeatureExtractor.forward (RefVSR_IR.py:502-546); x [n, 5, 3, h, w], h, w multiples of 4."""
    n, t, c, h, w = x.shape
    l1 = lrelu(conv(x.reshape(-1, c, h, w), W, P + 'conv_first'), 0.1)
    for i in range(5):                                                  # ResidualBlockNoBN x 5
        l1 = l1 + conv(F.relu(conv(l1, W, P + 'feature_extraction.%d.conv1' % i)), W, P + 'feature_extraction.%d.conv2' % i)
    cm = lambda z, name, stride=1: lrelu(conv(z, W, P + name + '.conv', stride=stride), 0.1)
    l2 = cm(cm(l1, 'feat_l2_conv1', 2), 'feat_l2_conv2')
    l3 = cm(cm(l2, 'feat_l3_conv1', 2), 'feat_l3_conv2')
    l1, l2, l3 = l1.view(n, t, M, h, w), l2.view(n, t, M, h // 2, w // 2), l3.view(n, t, M, h // 4, w // 4)
    ref = [l1[:, center], l2[:, center], l3[:, center]]
    aligned = torch.stack([pcd_alignment([l1[:, i], l2[:, i], l3[:, i]], ref, W, P + 'pcd_alignment.') for i in range(t)], 1)
    return tsa_fusion(aligned, W, P + 'fusion.', center)


class OracleNetworkIR(orc.OracleNetwork):
    """Stateful restatement of models/archs/RefVSR_IR.py:Network (inference).  Executes what the reference executes."""

    def __init__(self, config, W):
        orc.OracleNetwork.__init__(self, config, W)
        self.stride = config.keyframe_stride
        self.keyframe_idx = None

    def _refill(self, lrs, h, w):
        """spatial_padding + compute_refill_features (RefVSR_IR.py:171-217): reflect pad to a multiple of 4, temporal
        padding [4, 3 | 0..t-1 | -4, -5], EDVR on the 5-frame window around every keyframe."""
        n, t, c, hh, ww = lrs.shape
        ph, pw = (4 - hh % 4) % 4, (4 - ww % 4) % 4
        if ph or pw:
            iy = orc.reflect_index(torch.arange(0, hh + ph), hh)
            ix = orc.reflect_index(torch.arange(0, ww + pw), ww)
            lrs = lrs[:, :, :, iy][:, :, :, :, ix]
        ext = torch.cat([lrs[:, [4, 3]], lrs, lrs[:, [-4, -5]]], 1)
        return {int(i): edvr_features(ext[:, i:i + 5], self.W)[:, :, :h, :w] for i in self.keyframe_idx}

    def forward(self, lrs, refs, is_first_frame, is_log=False, is_train=False, trace=None):
        assert not is_train
        W, C = self.W, self.C
        n, t, c, h, w = lrs.shape
        ctr = t // 2
        assert h >= 64 and w >= 64 and t >= 5
        if self.max_frame_itr_num is not None and self.frame_itr_num == self.max_frame_itr_num:
            is_first_frame = True                                            # :239-241
        ff = [orc.spynet(lrs[:, j + 1], lrs[:, j], W) for j in range(t - 1)]    # :249-258
        bf = [orc.spynet(lrs[:, j - 1], lrs[:, j], W) for j in range(1, t)]
        if is_first_frame:                                                   # :262-272
            self.keyframe_idx = np.arange(0, t, self.stride)
        else:
            ki = self.keyframe_idx - 1
            ki = ki[ki >= 0]
            self.keyframe_idx = np.arange(ki[0], t, self.stride)
        if self.keyframe_idx[-1] != t - 1:
            self.keyframe_idx = np.append(self.keyframe_idx, t - 1)
        refill = self._refill(lrs, h, w)
        match = [self._feature_match(lrs[:, i], refs[:, i]) for i in range(t)]    # :279-285 (every frame, every call)
        conf_maps, index_maps = [m[0] for m in match], [m[1] for m in match]
        # ---- backward branch over ALL frames (:292-326)
        feat = torch.zeros(n, C, h, w)
        feat_up = torch.zeros(n, C, 2 * h, 2 * w)
        conf = torch.zeros(n, 1, h, w)
        outputs = [None] * t
        flow = None
        for i in range(t - 1, -1, -1):
            if i < t - 1:
                flow = bf[i]
                feat = orc.warp(feat, flow)
                conf = orc.warp(conf, flow)
                feat_up = orc.warp(feat_up, orc.flow_up2(flow))
            if i in self.keyframe_idx:
                feat = conv(torch.cat([feat, refill[i]], 1), W, 'Network.backward_fusion')
            rf, rfd = self._ref_feats(refs[:, i])
            x = orc.resblocks_with_input_conv(torch.cat([lrs[:, i], feat], 1), W, 'Network.backward_resblocks', self.nb)
            feat, feat_up, conf = self._rap(lrs[:, i], refs[:, i], conf_maps[i], conf, index_maps[i], x, feat_up, rfd, rf)
            if i == ctr:
                bw_up, conf_bw = feat_up, conf
            outputs[i] = feat
        # ---- forward branch over frames 0..ctr (:328-365).  NOTE the reference's quirks, reproduced: `flow` inside the loop
        # is the variable left over from the backward loop (= backward_flows[:, 0]) for the 2x map and the confidence map; the
        # loop starts at frame 0 in every call (the carried state is used instead of zeros when is_first_frame is False);
        # feat / feat_up / conf enter the loop with their values from the END of the backward loop unless is_first_frame.
        if is_first_frame:
            feat = torch.zeros_like(feat)
            feat_up = torch.zeros_like(bw_up)
            conf = torch.zeros_like(conf_maps[0])
        for i in range(0, ctr + 1):
            if i > 0:
                feat = orc.warp(feat, ff[i - 1])
                feat_up = orc.warp(feat, orc.flow_up2(flow))
                conf = orc.warp(conf, flow)
            elif not is_first_frame:
                feat = orc.warp(self.forward_feat_prop_prev, self.forward_flow_prev)
                feat_up = orc.warp(self.forward_feat_prop_UP_prev, orc.flow_up2(self.forward_flow_prev))
                conf = orc.warp(self.forward_conf_map_prop_prev, self.forward_flow_prev)
            if i in self.keyframe_idx:
                feat = conv(torch.cat([feat, refill[i]], 1), W, 'Network.forward_fusion')
            rf, rfd = self._ref_feats(refs[:, i])
            x = orc.resblocks_with_input_conv(torch.cat([lrs[:, i], outputs[i], feat], 1), W, 'Network.forward_resblocks', self.nb)
            feat, feat_up, conf = self._rap(lrs[:, i], refs[:, i], conf_maps[i], conf, index_maps[i], x, feat_up, rfd, rf)
            if i == 0:                                                       # :360-365
                self.forward_feat_prop_prev = feat.clone()
                self.forward_flow_prev = ff[0].clone()
                self.forward_feat_prop_UP_prev = feat_up.clone()
                self.forward_conf_map_prop_prev = conf.clone()
        base = orc.bicubic_scale(lrs[:, ctr], self.scale, clamp=True)
        out = self._compute_up(bw_up, feat_up, conf_bw, conf, base)
        if is_first_frame:
            self.frame_itr_num = 0
        self.frame_itr_num += 1
        outs = collections.OrderedDict()
        if is_log:                                                           # :229-230
            outs['vis'] = collections.OrderedDict()
            if self.cfg.save_sample:                                         # :367-384 (conf / index maps of frame t//2)
                outs['vis'].update(self.sample_vis(lrs[:, ctr], refs[:, ctr], index_maps[ctr], conf_maps[ctr], conf_bw, conf))
        outs['result'] = out.clamp(0, 1)
        if trace is not None:
            trace.update(keyframe_idx=self.keyframe_idx.copy(), refill=refill, is_first_frame=is_first_frame)
        return outs

    __call__ = forward
