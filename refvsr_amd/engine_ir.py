"""RefVSR_IR on the MI355X kernels: `Network.forward` of /root/reference/models/archs/RefVSR_IR.py (the IconVSR-style
variant: information refill from an EDVR-M feature extractor, PCD alignment on modulated deformable convolutions, TSA
fusion -- models/archs/edvr_net.py), one sample at a time, as launches of the HIP kernels behind the C-ABI.

Shares everything RefVSR_IR shares with RefVSR (SPyNet, matching, reference encoders, alignment, RAP fusion, upsampler,
the per-frame window cache) with refvsr_amd/engine.py; adds the EDVR extractor (its per-frame feature pyramid is cached
per frame like the matching) and the different propagation order: backward branch over ALL t frames, forward branch
restarted at frame 0 of every window from the carried state, key-frame refill -- including the reference's quirks (the
forward branch warps its 2x map and confidence with the flow variable left over from the backward loop, :335-337).
Feature maps have C = 36 channels (channel stride 40, zero padding).  Sequential on the caller's stream.
"""
import collections

import numpy as np
import torch

from . import ops
from .engine import Engine, Weights
from .packing import pack_conv

M = 64          # EDVR-M channels
DG = 8          # deformable groups


class WeightsIR(Weights):
    def __init__(self, config, sd, device):
        Weights.__init__(self, config, sd, device)
        C = self.C
        g = lambda n: (sd['Network.' + n + '.weight'], sd['Network.' + n + '.bias'])

        def mf(name, srcs):
            w, b = g(name)
            self.conv[name] = ops.ConvWeights(pack_conv(w, b, srcs), device)

        mf('backward_fusion', [C, M])
        mf('forward_fusion', [C, M])
        E = 'edvr.'
        mf(E + 'conv_first', [3])
        for i in range(5):
            mf(E + 'feature_extraction.%d.conv1' % i, [M])
            mf(E + 'feature_extraction.%d.conv2' % i, [M])
        for nm in ('feat_l2_conv1', 'feat_l2_conv2', 'feat_l3_conv1', 'feat_l3_conv2'):
            mf(E + nm + '.conv', [M])
        A = E + 'pcd_alignment.'
        for lv in ('l3', 'l2', 'l1'):
            mf(A + 'offset_conv1.%s.conv' % lv, [M, M])
        mf(A + 'offset_conv2.l3.conv', [M])
        mf(A + 'offset_conv2.l2.conv', [M, M])
        mf(A + 'offset_conv2.l1.conv', [M, M])
        mf(A + 'offset_conv3.l2.conv', [M])
        mf(A + 'offset_conv3.l1.conv', [M])
        mf(A + 'feat_conv.l2.conv', [M, M])
        mf(A + 'feat_conv.l1.conv', [M, M])
        mf(A + 'cas_offset_conv1.conv', [M, M])
        mf(A + 'cas_offset_conv2.conv', [M])
        for nm in ('dcn_pack.l3', 'dcn_pack.l2', 'dcn_pack.l1', 'cas_dcnpack'):
            mf(A + nm + '.conv_offset', [M])
            # the DCN's contraction as a 1x1 conv over the sampled columns, column channel = tap * 64 + channel
            w, b = g(A + nm)
            self.conv[A + nm] = ops.ConvWeights(pack_conv(w.permute(0, 2, 3, 1).reshape(w.shape[0], 9 * M, 1, 1), b, [9 * M]), device)
        F_ = E + 'fusion.'
        mf(F_ + 'temporal_attn1', [M])
        mf(F_ + 'temporal_attn2', [M])
        mf(F_ + 'feat_fusion.conv', [5 * M])
        mf(F_ + 'spatial_attn1.conv', [5 * M])
        mf(F_ + 'spatial_attn2.conv', [2 * M])
        mf(F_ + 'spatial_attn3.conv', [M])
        mf(F_ + 'spatial_attn4.conv', [M])
        mf(F_ + 'spatial_attn5', [M])
        mf(F_ + 'spatial_attn_l1.conv', [M])
        mf(F_ + 'spatial_attn_l2.conv', [2 * M])
        mf(F_ + 'spatial_attn_l3.conv', [M])
        mf(F_ + 'spatial_attn_add1.conv', [M])
        mf(F_ + 'spatial_attn_add2', [M])


class EngineIR(Engine):
    split_state_ok = False               # its branches read fw_feat_up directly: one-message hand-off only

    def __init__(self, config, weights):
        Engine.__init__(self, config, weights)
        self.stride = config.keyframe_stride
        self.keyframe_idx = None
        self.Cs = (self.C + 7) // 8 * 8

    def reset_state(self):
        Engine.reset_state(self)
        self.keyframe_idx = None

    def group_ok(self):
        """Frame groups, phase-A groups and the n > 1 multi-map path are schedules of RefVSR's propagation (Engine._prop_step_b / rap_b):
        an IR engine never takes them, whatever its mid_channels."""
        return False

    def set_pipelined(self, on=True):
        """Cross-call pipelining (round 4; Engine.set_pipelined's contract): everything that is a function of the window's frames
        only -- matching and alignment of the new frame, its EDVR pyramid features, the flows, the refill features of the key frames
        (PCD alignment + TSA fusion, RefVSR_IR.py:193-217) -- runs on the preparation stream while the other stream is still
        walking the previous call's two propagation branches."""
        self.pipelined = bool(on)

    # ------------------------------------------------------------------ EDVR-M feature extractor
    def _pyramid_feats(self, fr, ph, pw):
        """L1 / L2 / L3 features of one frame (RefVSR_IR.py:514-520), cached on the frame context."""
        if getattr(fr, 'edvr', None) is None:
            E = 'edvr.'
            lr = fr.lr
            if ph or pw:                     # spatial_padding (:171-191): reflect pad to a multiple of 4 (torch: data movement)
                lr = torch.nn.functional.pad(lr[None], [0, pw, 0, ph], mode='reflect')[0].contiguous()
                x8 = ops.pack_nhwc16(lr, 8)
            else:
                x8 = fr.lr8 if fr.lr8 is not None else ops.pack_nhwc16(lr, 8)
            l1 = ops.conv(self.cw(E + 'conv_first'), x8, act=0.1)
            pairs = [(self.cw(E + 'feature_extraction.%d.conv1' % i), self.cw(E + 'feature_extraction.%d.conv2' % i)) for i in range(5)]
            for c1, c2 in pairs:             # ResidualBlockNoBN x 5 (64 channels: two launches per block)
                l1 = ops.conv(c2, ops.conv(c1, l1, act=0.0), res=l1)
            l2 = ops.conv(self.cw(E + 'feat_l2_conv2.conv'), ops.conv(self.cw(E + 'feat_l2_conv1.conv'), l1, stride=2, act=0.1), act=0.1)
            l3 = ops.conv(self.cw(E + 'feat_l3_conv2.conv'), ops.conv(self.cw(E + 'feat_l3_conv1.conv'), l2, stride=2, act=0.1), act=0.1)
            fr.edvr = (l1, l2, l3)
        return fr.edvr

    def _dcn(self, x, extra, name, act):
        """ModulatedDCNPack (edvr_net.py:49-56): conv_offset -> sampling -> 1x1 contraction (+ LeakyReLU 0.1 where the caller applies one)."""
        om = ops.conv(self.cw(name + '.conv_offset'), extra, planar_out=True)
        return ops.conv(self.cw(name), ops.dcn_sample(x, om, DG), act=act)

    def _pcd(self, nbr, ref):
        """PCDAlignment.forward (edvr_net.py:134-185)."""
        A = 'edvr.pcd_alignment.'
        cm = lambda name, a, b=None, act=0.1: ops.conv(self.cw(A + name + '.conv'), a, b, act=act)
        up_off = up_feat = feat = None
        for i in (3, 2, 1):
            lv = 'l%d' % i
            off = cm('offset_conv1.' + lv, nbr[i - 1], ref[i - 1])
            if i == 3:
                off = cm('offset_conv2.' + lv, off)
            else:
                off = cm('offset_conv3.' + lv, cm('offset_conv2.' + lv, off, up_off))
            feat = self._dcn(nbr[i - 1], off, A + 'dcn_pack.' + lv, 0.1 if i == 3 else 1.0)
            if i < 3:
                feat = cm('feat_conv.' + lv, feat, up_feat, act=0.1 if i == 2 else 1.0)
            if i > 1:
                up_off = ops.up2_bilinear_nhwc16(off, 2.0)
                up_feat = ops.up2_bilinear_nhwc16(feat)
        off = cm('cas_offset_conv2', cm('cas_offset_conv1', feat, ref[0]))
        return self._dcn(feat, off, A + 'cas_dcnpack', 0.1)

    def _tsa(self, aligned, center):
        """TSAFusion.forward (edvr_net.py:248-300)."""
        P = 'edvr.fusion.'
        cm = lambda name, a, act=0.1, res=None: ops.conv(self.cw(P + name), a, act=act, res=res)
        emb_ref = cm('temporal_attn1', aligned[center], act=1.0)
        emb = [cm('temporal_attn2', a, act=1.0) for a in aligned]
        al = ops.tsa_weight(aligned, emb, emb_ref)
        feat = cm('feat_fusion.conv', al)
        attn = cm('spatial_attn2.conv', ops.pool3s2_pair(cm('spatial_attn1.conv', al)))
        lvl = cm('spatial_attn_l2.conv', ops.pool3s2_pair(cm('spatial_attn_l1.conv', attn)))
        lvl = ops.up2_bilinear_nhwc16(cm('spatial_attn_l3.conv', lvl))
        attn = cm('spatial_attn3.conv', attn, res=lvl)
        attn = ops.up2_bilinear_nhwc16(cm('spatial_attn4.conv', attn))
        attn = cm('spatial_attn5', attn, act=1.0)
        attn_add = cm('spatial_attn_add2', cm('spatial_attn_add1.conv', attn), act=1.0)
        return ops.tsa_blend(feat, attn, attn_add)

    def _refill(self, fr, t, h, w):
        """compute_refill_features (RefVSR_IR.py:193-217): temporal padding [4, 3 | 0..t-1 | t-4, t-5], EDVR on the 5-frame
        window around every key frame."""
        ph, pw = (4 - h % 4) % 4, (4 - w % 4) % 4
        ext = [4, 3] + list(range(t)) + [t - 4, t - 5]
        out = {}
        for k in self.keyframe_idx:
            k = int(k)
            feats = [self._pyramid_feats(fr[ext[k + j]], ph, pw) for j in range(5)]
            aligned = [self._pcd(list(f), list(feats[2])) for f in feats]
            r = self._tsa(aligned, 2)
            if ph or pw:
                r = r[:h, :w].contiguous()
            out[k] = r
        return out

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, lrs, refs, is_first_frame, want_vis=False, frame_ids=None, want_log=False, input_ready=None):
        if self.takes_pipelined_path(frame_ids, want_log) and not want_vis:
            with torch.cuda.device(lrs.device):
                return self._forward_ir_pipelined(lrs, refs, is_first_frame, frame_ids, input_ready), None
        with torch.cuda.device(lrs.device), ops.on_stream(torch.cuda.current_stream()):
            if isinstance(input_ready, torch.cuda.Stream):
                torch.cuda.current_stream().wait_stream(input_ready)
            elif isinstance(input_ready, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(input_ready)
            out, dbg = self._forward_ir(lrs, refs, is_first_frame, frame_ids, bool(want_log and want_vis))
            if self._pipe is not None:           # a sequential call (is_log) between pipelined ones: the internal streams see its state
                for st in set(self._pipe):
                    st.wait_stream(torch.cuda.current_stream())
        # RefVSR_IR returns no 'eval_vis' (RefVSR_IR.py:366-386); `vis` holds the save_sample block only (:374-384)
        return out, ((None, dbg) if want_log else None)

    @torch.no_grad()
    def _forward_ir_pipelined(self, lrs, refs, is_first_frame, frame_ids, input_ready=None):
        """_forward_ir over two internal streams (Engine._forward_pipelined's rules: dependencies by HIP events, every tensor that
        crosses streams recorded on its consumers, the host at most pipe_depth calls ahead): P = _ir_part_a of this call, M = the
        previous call's _ir_part_b, then this one's.  Restarts of the forward branch run both parts on M, after everything in
        flight.  Results are bit-identical to the sequential call."""
        dev = lrs.device
        caller = torch.cuda.current_stream()
        M, _, _, P = self._pipe_streams(dev)
        while len(self._inflight) >= self.pipe_depth:
            self._inflight.popleft().synchronize()
        if input_ready is None:
            input_ready = torch.cuda.Event()
            input_ready.record(caller)
        if not isinstance(input_ready, str):
            for st in (M, P):
                if isinstance(input_ready, torch.cuda.Stream):
                    st.wait_stream(input_ready)
                else:
                    st.wait_event(input_ready)
        elif input_ready != 'materialised':
            raise ValueError("input_ready must be None, 'materialised', a torch.cuda.Event or a torch.cuda.Stream")
        for st in (M, P):
            lrs.record_stream(st)
            refs.record_stream(st)
        restart = is_first_frame or self.fw_feat is None or (self.max_frame_itr_num is not None and self.frame_itr_num == self.max_frame_itr_num)
        if restart:
            M.wait_stream(P)
            with ops.on_stream(M):
                out, _ = self._forward_ir(lrs, refs, is_first_frame, frame_ids)
            P.wait_stream(M)
        else:
            with ops.on_stream(P):
                pa = self._ir_part_a(lrs, refs, False, frame_ids, share=(M, P))
                ev_p = torch.cuda.Event()
                ev_p.record()
            with ops.on_stream(M):
                M.wait_event(ev_p)
                out, _ = self._ir_part_b(pa)
        done = torch.cuda.Event()
        done.record(M)
        caller.wait_event(done)
        out.record_stream(caller)
        self._inflight.append(done)
        return out

    def _forward_ir(self, lrs, refs, is_first_frame, frame_ids, sample=False):
        pa = self._ir_part_a(lrs, refs, is_first_frame, frame_ids)
        return self._ir_part_b(pa, sample)

    def _ir_part_a(self, lrs, refs, is_first_frame, frame_ids, share=None):
        """Everything of a call that is a function of the window's frames (and of the key-frame schedule, a host-side counter):
        per-frame preparation, the refill features of the key frames, every flow the branches will ask for.  share: the streams
        that will consume the results besides the current one (pipelined mode)."""
        t, h, w = self._check_window(lrs, refs)
        if h < 64 or w < 64 or t < 5:
            raise RuntimeError('RefVSR_IR needs frames of at least 64x64 and a window of at least 5 frames (RefVSR_IR.py:244-246,203)')
        ctr = t // 2
        if self.max_frame_itr_num is not None and self.frame_itr_num == self.max_frame_itr_num:
            is_first_frame = True
        if not is_first_frame and self.fw_feat is None:
            raise RuntimeError('is_first_frame=False but no forward state is held (first call of a stream must pass is_first_frame=True)')
        if is_first_frame and frame_ids is not None:
            self.id_cache, self.flow_cache = {}, {}
        fr = self._frames(lrs, refs, frame_ids)
        if is_first_frame:                                                   # :262-272
            self.keyframe_idx = np.arange(0, t, self.stride)
        else:
            ki = self.keyframe_idx - 1
            ki = ki[ki >= 0]
            self.keyframe_idx = np.arange(ki[0], t, self.stride)
        if self.keyframe_idx[-1] != t - 1:
            self.keyframe_idx = np.append(self.keyframe_idx, t - 1)
        keys = set(int(k) for k in self.keyframe_idx)
        for f in fr:
            self.prepare_frame(f)                                            # matching of EVERY frame (:279-285), cached per frame
        refill = self._refill(fr, t, h, w)
        if share:
            # the flows of both branches in batched SPyNet passes, each carrying an event for its consumers
            self.flows([(fr[i], fr[i + 1]) for i in range(t - 1)] + [(fr[i], fr[i - 1]) for i in range(1, ctr + 1)], share)
            for f in fr:
                for x in [f.lr, f.ref, f.lr8, f.conf, f.idx, f.aligned, f.aligned_up] + list(f.pyr or []) + list(getattr(f, 'edvr', None) or []):
                    for st in share:
                        x.record_stream(st)
            for x in refill.values():
                for st in share:
                    x.record_stream(st)
        return dict(fr=fr, t=t, h=h, w=w, keys=keys, refill=refill, is_first_frame=is_first_frame, share=share)

    def _ir_part_b(self, pa, sample=False):
        """The two propagation branches and the upsampler (RefVSR_IR.py:292-365): the part that carries the state."""
        fr, t, h, w, keys, refill, is_first_frame, share = (pa[k] for k in ('fr', 't', 'h', 'w', 'keys', 'refill', 'is_first_frame', 'share'))
        ctr, dev, Cs = t // 2, fr[0].lr.device, self.Cs
        flow = lambda a, b: self.flow(fr[a], fr[b], share)
        zeros = lambda hh, ww: torch.zeros((hh, ww, Cs), dtype=torch.float16, device=dev)
        # ---- backward branch over all frames (:292-326)
        feat, feat_up = zeros(h, w), zeros(2 * h, 2 * w)
        conf = torch.zeros((1, h, w), dtype=torch.float32, device=dev)
        outputs = [None] * t
        fl = None
        for i in range(t - 1, -1, -1):
            if i < t - 1:
                fl = flow(i, i + 1)
                feat = ops.warp_nhwc16(feat, fl)
                conf = ops.warp_planar(conf, fl)
                feat_up = ops.warp_nhwc16(feat_up, ops.flow_up2(fl))
            if i in keys:
                feat = ops.conv(self.cw('backward_fusion'), feat, refill[i])
            x = self.resblocks(fr[i].lr8, feat, 'backward_resblocks')
            feat, feat_up, conf = self.rap(fr[i], conf, x, feat_up)
            if i == ctr:
                bw_up, conf_bw = feat_up, conf
            outputs[i] = feat
        # ---- forward branch, frames 0..ctr (:328-365; `fl` is the backward loop's last flow, as in the reference)
        if is_first_frame:
            feat, feat_up = zeros(h, w), zeros(2 * h, 2 * w)
            conf = torch.zeros((1, h, w), dtype=torch.float32, device=dev)
        new_state = None
        for i in range(0, ctr + 1):
            if i > 0:
                feat = ops.warp_nhwc16(feat, flow(i, i - 1))
                feat_up = ops.warp_nhwc16(feat, ops.flow_up2(fl))
                conf = ops.warp_planar(conf, fl)
            elif not is_first_frame:
                feat = ops.warp_nhwc16(self.fw_feat, self.fw_flow)
                feat_up = ops.warp_nhwc16(self.fw_feat_up, ops.flow_up2(self.fw_flow))
                conf = ops.warp_planar(self.fw_conf, self.fw_flow)
            if i in keys:
                feat = ops.conv(self.cw('forward_fusion'), feat, refill[i])
            lr_bw = torch.cat([fr[i].lr8, outputs[i]], 2)                    # [lr | backward features] as one HWC map (plumbing copy)
            x = ops.conv(self.cw('forward_resblocks.main.0'), lr_bw, feat, act=0.1)
            pairs = [(self.cw('forward_resblocks.main.2.%d.conv1' % j), self.cw('forward_resblocks.main.2.%d.conv2' % j)) for j in range(self.nb)]
            x = self._block_chain(x, pairs, 0.0)
            feat, feat_up, conf = self.rap(fr[i], conf, x, feat_up)
            if i == 0:
                new_state = (feat, feat_up, conf, flow(1, 0))
        self.fw_feat, self.fw_feat_up, self.fw_conf, self.fw_flow = new_state
        out = self.compute_up(bw_up, feat_up, conf_bw, conf, fr[ctr].lr)
        if is_first_frame:
            self.frame_itr_num = 0
        self.frame_itr_num += 1
        dbg = self._sample_vis(fr[ctr], conf_bw, conf) if sample else collections.OrderedDict()
        return out, dbg
