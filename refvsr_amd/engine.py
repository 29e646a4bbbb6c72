"""The MI355X RefVSR pipeline: `Network.forward` of the reference re-expressed as launches of the
HIP kernels behind the C-ABI (refvsr_amd/ops.py), one sample (n=1) at a time.

Follows /root/reference/models/archs/RefVSR.py:151-325 step by step (the comments cite the lines).
What differs from the reference's *execution* (never its results):

  * only the flows the network consumes are computed (RefVSR.py:182-191 computes 2(t-1), uses
    t//2 + 1 in steady state), and
  * everything that depends on a single frame only -- SPyNet pyramids, matching (conf, index),
    the reference encoders, and both AlignedAttention outputs (aa1 gather, aa2 gather +
    AlignedConv2d), which are functions of (lr_i, ref_i) alone -- is cached per frame and reused
    across overlapping sliding windows and across the backward/forward branches
    (`config.cache_windows`).  Frames are recognised by exact content
    equality with the previous window, so the drop-in call signature needs no frame ids.

Feature maps live in HBM as fp16 HWC ("nhwc16"); frames, flows and confidence maps as planar fp32.
"""
import collections
import itertools
import os

import torch

from . import ops
from .knobs import env_flag
from .packing import pack_conv
from .weights import VGG_MEAN, VGG_STD

_uid = itertools.count()


def _timed_event(stream):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream)
    return e


class FrameCtx(object):
    """Per-frame data (functions of one (lr, ref) frame pair only)."""

    def __init__(self, lr, ref):
        self.uid = next(_uid)
        # the context OWNS its frames (clones): it outlives the call in the window cache, and a caller that refills one
        # static input buffer in place must neither change cached data nor make the content compare see "equal" frames
        self.lr = lr.clone()    # planar fp32 [3,h,w]
        self.ref = ref.clone()
        self.lr8 = None         # nhwc16 [h,w,8]
        self.pyr = None         # SPyNet pyramid of lr, coarse -> fine
        self.conf = None        # planar [1,h,w]
        self.idx = None         # int32 [Hf*Wf]
        self.aligned = None     # aa1 output nhwc16 [h,w,C]
        self.aligned_up = None  # aa2 output nhwc16 [2h,2w,C]
        self.bw_head = None     # pipelined mode: (n, map) -- the first n layers of the backward branch's first step on this frame
        self.ready = None       # pipelined mode: HIP event recorded on the preparation stream once all of the above exist


class Weights(object):
    """All parameters of the network packed for the kernels (built from a reference-keyed state dict)."""

    def __init__(self, config, sd, device):
        self.device = device
        self.C = config.mid_channels
        self.nb = config.num_blocks
        self.conv = {}
        self.raw = {}
        self.chains = {}            # runs of fused blocks -> ops.ResblockChain pointer tables (Engine._block_chain)
        C = self.C
        g = lambda n: (sd['Network.' + n + '.weight'], sd['Network.' + n + '.bias'])

        def mf(name, srcs, shuffle=False, mt=None):
            w, b = g(name)
            self.conv[name] = ops.ConvWeights(pack_conv(w, b, srcs, shuffle, mt=mt), device)

        def dr(name):
            w, b = g(name)
            self.raw[name] = (w.detach().to(device, torch.float32).contiguous(),
                              b.detach().to(device, torch.float32).contiguous())

        def mf32(name, srcs):
            w, b = g(name)
            self.conv[name] = ops.ConvWeights(pack_conv(w, b, srcs, f32=True), device)

        chans = [(32, 8), (64, 32), (32, 64), (16, 32), (2, 16)]
        # SPyNet's streamed 7x7 convs (every conv but the first of a level) on plain fp16 weights: half the weight stream and
        # half the MFMAs of the kernels that are bound by both.  Their share of the end-to-end error budget is measured
        # (tools/lo_term_study.py); config.spynet_hi_lo = True / REFVSR_SPYNET_HILO=1 restores hi + lo.
        hi_only = not (bool(getattr(config, 'spynet_hi_lo', False)) or env_flag('REFVSR_SPYNET_HILO'))
        for lvl in range(6):
            for j, (co, ci) in enumerate(chans):
                name = 'FlowNet.basic_module.%d.basic_module.%d.conv' % (lvl, j)
                w, b = g(name)
                self.conv[name] = ops.ConvWeights(pack_conv(w, b, [ci], hi_only=hi_only and j > 0), device)
                if co > 16 and ci > 8:
                    # the streamed 7x7 convs of the coarse pyramid levels (a handful of pixel tiles): 16 output channels per
                    # workgroup, so that 2-4x as many CUs share the 200-400 KB weight stream (Engine.flow picks per level)
                    self.conv[name + '/mt1'] = ops.ConvWeights(pack_conv(w, b, [ci], mt=1, hi_only=hi_only), device)
        fe = 'feature_match.feature_extract.'
        self.hd = bool(config.flag_HD_in)
        self.vgg7 = self.hd or config.scale != 4          # attention.py:31-35
        dr('feature_match.sub_mean')
        mf32(fe + '0', [3])
        mf32(fe + '2', [64])
        if self.vgg7:                     # VGG19[0:7] + map128 (attention.py:33-40)
            mf32(fe + '5', [64])
            mf32(fe + 'map128.0', [128])
            self.raw[fe + 'map128.0'] = tuple(t.detach().to(device, torch.float32).contiguous() for t in
                                              (g(fe + 'map128.0')[0].reshape(16, 128), g(fe + 'map128.0')[1]))
        else:
            mf32(fe + 'map64.0', [64])
            self.raw[fe + 'map64.0'] = tuple(t.detach().to(device, torch.float32).contiguous() for t in
                                             (g(fe + 'map64.0')[0].reshape(16, 64), g(fe + 'map64.0')[1]))

        def aligned(prefix, stride):
            mf(prefix + '.conv1.0', [3])
            mf(prefix + '.conv1.2.conv1', [32])
            mf(prefix + '.conv1.2.conv2', [32])
            # the strided 5x5 predictor conv (alignment.py:20: 64 -> 32, stride = the patch size): with 32 output channels per workgroup
            # its staged input tile + a weight ring exceeds the LDS and the generic kernel runs in gather mode (B fragments from global
            # memory inside the K loop); at stride 2 sixteen output channels per workgroup (mt = 1) fit the 4 x 32 tile mode: 116.6 ->
            # 94.9 us at 540 x 960, bit-identical (profiles/r05_pconv_mt1_ab.txt; at stride 4 no tile fits either way and mt = 1 only
            # doubles the gathers: 196 vs 421 us at 1080 x 1920).  REFVSR_PCONV_MT = 1 | 2 forces one packing (A/B knob)
            mt_p = int(os.environ.get('REFVSR_PCONV_MT', '1' if stride == 2 else '2'))
            mf(prefix + '.p_conv.0', [32, 32], mt=mt_p)
            mf(prefix + '.p_conv.2.conv1', [32])
            mf(prefix + '.p_conv.2.conv2', [32])
            mf(prefix + '.p_conv.4', [32])
        aligned('aa2.align', config.matching_ksize)
        if config.matching_ksize // 2 > 1:     # RefVSR.py:39 -- aa1 aligns only when its patch is larger than 1 px
            aligned('aa1.align', config.matching_ksize // 2)

        def reslist(name, n):
            for i in range(n):
                mf('%s.RBs.%d.conv1' % (name, i), [C])
                mf('%s.RBs.%d.conv2' % (name, i), [C])
            mf(name + '.conv_tail', [C])

        mf('ref_encoder1.0.0', [3])
        mf('ref_encoder1.1.0', [C])
        reslist('res1', 4)
        mf('ref_encoder2.0.0', [C])
        mf('ref_encoder2.1.0', [C])
        reslist('res2', 4)
        for nm in ('conf_fusion', 'conf_fusion2', 'conf_fusion_BWFW'):
            dr(nm + '.0.0')
            mf(nm + '.1.0', [16])
        for nm in ('feat_fusion', 'feat_fusion2', 'feat_fusion_BWFW'):
            mf(nm + '.0.0', [C, C])
            mf(nm + '.1.0', [C])
        mf('feat_fusion2_1.0.0', [C, C])
        reslist('feat_decoder', 8)
        reslist('feat_decoder2', 4)
        reslist('feat_decoder_BWFW', 4)
        for br in ('backward_resblocks', 'forward_resblocks'):
            cin = sd['Network.' + br + '.main.0.weight'].shape[1]
            if cin == C + 3:
                mf(br + '.main.0', [3, C])
            else:                        # RefVSR_IR forward branch: cat([lr, backward features, state]) (RefVSR_IR.py:354); the
                assert cin == 2 * C + 3  # kernel takes two maps, so [lr | backward features] are handed over as one HWC map
                mf(br + '.main.0', [3, C, C])
                cw = self.conv[br + '.main.0']
                cw.cpads = [cw.cpads[0] + cw.cpads[1], cw.cpads[2]]
            for i in range(self.nb):
                mf('%s.main.2.%d.conv1' % (br, i), [C])
                mf('%s.main.2.%d.conv2' % (br, i), [C])
        mf('fusion_UP', [C, C])
        mf('upsample1.upsample_conv', [C], shuffle=True)
        if config.scale == 4:                             # RefVSR.py:89-90
            mf('upsample2.upsample_conv', [C], shuffle=True)
        mf('conv_hr', [C])
        mf('conv_last', [C])


class Engine(object):
    """Stateful per-stream executor (one stream of consecutive frames, like a reference module
    instance: RefVSR.py:96-101,279-283)."""

    def __init__(self, config, weights):
        if config.scale not in (2, 4):
            raise NotImplementedError('SR scale must be 2 or 4 (configs/config_RefVSR_*.py: "SR scale (2 | 4)")')
        self.cfg = config
        self.W = weights
        self.C = config.mid_channels
        self.nb = config.num_blocks
        self.ks = config.matching_ksize
        self.hd = bool(config.flag_HD_in)
        self.vgg7 = self.hd or config.scale != 4           # matching on VGG19[0:7] features: grid = half the (resized) frame
        self.cache = bool(getattr(config, 'cache_windows', True))
        self.match_row_splits = 1
        self.match_margin = float(getattr(config, 'match_exact_margin', ops.MATCH_EXACT_MARGIN))
        self.chain_calls = bool(getattr(config, 'chain_calls', not env_flag('REFVSR_NO_CHAIN_CALLS')))
        self.fuse_resblocks = (bool(getattr(config, 'fuse_resblocks', True)) and ops.resblock_fits(self.C)
                               and not env_flag('REFVSR_NO_FUSE'))
        self.rb24 = not env_flag('REFVSR_NO_RB24')               # A/B knob: the generic lean kernel for C = 24 as well
        self.rb48 = (bool(getattr(config, 'fuse_resblocks', True)) and not env_flag('REFVSR_NO_FUSE')
                     and not env_flag('REFVSR_NO_RB48'))                 # A/B knob: C = 48 blocks as two refvsr_conv48 launches (round 3)
        self.rb48_max_pixels = int(os.environ.get('REFVSR_RB48_MAX_PIXELS', str(540 * 960)))
        self.rb48_multimap = not env_flag('REFVSR_NO_RB48_MULTIMAP')      # A/B knob: frame groups of C = 48 run their blocks map by map
        # SPyNet levels up to this many pixels run their streamed convs with 16 output channels per workgroup (A/B knob; 0 = never)
        self.spynet_mt1_pixels = int(os.environ.get('REFVSR_SPYNET_MT1_PIXELS', str(72 * 120)))
        # pipelined mode: the backward branch restarts from zeros at the window's LAST frame (RefVSR.py:211-214), so the first
        # layers of that step are a function of that frame alone: its input conv + the first `bw_head_blocks` residual blocks run
        # with the frame's preparation on the P stream -- load balance between the two internal streams (results identical).
        # Measured (profiles/r04_knobs_ab.txt, same box, two rounds of five passes each): off 199.2 / 199.2 frames/s, 0 blocks 200.0 /
        # 199.5, 12 blocks 205.8 / 206.2, all 24 blocks 203.6 / 203.6 -> 12 (M 4.66 ms per call, P + F 4.63).  -1 = off.
        self.bw_head_blocks = min(self.nb, max(-1, int(getattr(config, 'bw_head_blocks', None) if getattr(config, 'bw_head_blocks', None) is not None
                                                         else os.environ.get('REFVSR_BW_HEAD_BLOCKS', '12'))))
        self.warp_up2 = not env_flag('REFVSR_NO_WARP_UP2')          # A/B knob: flow_up2 as its own launch + 2x flow map (round 3)
        self.fuse_head = not env_flag('REFVSR_NO_FUSE_HEAD')        # A/B knob: bicubic base map + generic planar conv (round 3)
        # conv_hr + head as ONE launch on the fused block's skeleton (refvsr_conv_hr_last): bit-identical to the two launches, 112 vs
        # 132 us stand-alone at 1080p, but NEUTRAL in the frame (209.3 / 209.9 vs 210.1 / 210.1 frames/s, profiles/
        # r04_tail_and_mfid_run.log: the HR intermediate it saves is 6 % of a launch pair that a 19 % larger conv1 pays for) -> opt-in
        self.fuse_tail = self.fuse_head and (env_flag('REFVSR_FUSE_TAIL') or bool(getattr(config, 'fuse_tail', False)))
        self.fuse_conf = not env_flag('REFVSR_NO_FUSE_CONF')        # A/B knob: confidence fusions as separate launches (round 3)
        # the 1x1 map that ends the matching's feature extractor on its own HBM-bound kernel (refvsr_conv1x1_f32) instead of the
        # generic conv's fp32 mode (118 us at 270 x 480 for 33 MB of reads); A/B knob REFVSR_NO_MAP1X1=1
        self.map1x1 = not env_flag('REFVSR_NO_MAP1X1')
        self.spynet_batch = not env_flag('REFVSR_NO_SPYNET_BATCH')   # A/B knob: one SPyNet pass per flow, as in round 3
        self.overlap = bool(getattr(config, 'overlap_streams', True)) and not env_flag('REFVSR_NO_OVERLAP')
        self._side = None
        # pipelined mode (opt-in, see forward()): internal streams M (backward branch + upsampler), F (forward branch),
        # P (per-frame preparation + flows); the caller's stream only receives the result
        self.pipelined = bool(getattr(config, 'pipelined', False))
        self._zero_maps = {}
        # result format (round 6, extension; default 'float32' = the reference's): 'float16' | 'uint8' -- the output head stores
        # rint(255 v) itself (REFVSR_RESULT_U8: the bytes eval_qual_quan.py:117-119 writes to the PNG), a quarter of the bytes to copy
        self.result_dtype = str(getattr(config, 'result_dtype', None) or 'float32').replace('torch.', '')
        ops.result_format(self.result_dtype)
        self._pipe = None
        # stream layout of the pipelined mode when nothing is configured: 'pf_m' for one forward() per frame (round 4); an engine that
        # is driven through forward_group switches to 'pfm' (P | F | M) at its first group: with the backward branches batched the
        # P + F stream is the critical path (P 2.9 + F 1.9 ms per frame = the frame, M idle a third of the time); with F on its own
        # stream the forward-branch chain runs under the preparation of the group's new frames -- same box, G = 4: 213.1 (pf_m) /
        # 216.9 (p_fm) / 227.5 (pfm) frames/s, 228.9 with the backward head on M as well (profiles/r05_group_layout_ab.txt)
        self._layout_default = None               # None: by model (see _pipe_streams); forward_group / forward_multi set 'pfm'
        self.pipe_depth = max(1, int(getattr(config, 'pipe_depth', None) or os.environ.get('REFVSR_PIPE_DEPTH') or 3))
        self._inflight = collections.deque()
        self.stream_events = None      # bench.py: list collecting per-call section events of the pipelined mode's streams
        self.kernel_events = None      # bench.py: list collecting (start, end) HIP events of match_top2 launches
        self.chain_events = None       # bench.py: list collecting (start, end, blocks, h, w) of the fused-ResBlock runs
        self.reset_state()

    # ------------------------------------------------------------------ state
    def reset_state(self):
        self.fw_feat = None
        self.fw_flow = None
        self.fw_feat_up = None
        self.fw_conf = None
        self.frame_itr_num = 0
        self.max_frame_itr_num = self.cfg.reset_branch
        self.prev_window = []
        self.flow_cache = {}
        self.id_cache = {}
        self.ctx_pinned = {}
        self.ctx_strict = False
        self._fw_up_wait = None              # split hand-off: makes the current stream wait for fw_feat_up's arrival (import_state_head)

    def set_pipelined(self, on=True):
        """Opt in to cross-call pipelining: forward() is then given `frame_ids` (one hashable id per window frame; equal ids
        <=> equal (lr, ref) content within a stream) and spreads consecutive calls over the engine's internal streams.
        Inputs: forward(..., input_ready=) says when lrs / refs are final -- None (default): the internal streams wait for the
        caller's stream as it stands at the call (always safe; since that stream also waits for the previous result, calls then
        overlap only internally); 'materialised': the caller asserts the tensors are final (pre-loaded clip; what bench.py's
        headline does); an event / stream: the producer's, e.g. a copy stream (shard.EngineExecutor).  Results are bit-identical
        to the default mode in every case."""
        self.pipelined = bool(on)

    def export_state(self):
        """Forward-branch state as planar fp32 tensors (what RefVSR.py:279-283 keeps; RefVSR_IR adds its key-frame indices,
        RefVSR_IR.py:262-272)."""
        if self.fw_feat is None:
            return None
        self._await_fw_up(self.fw_feat_up)
        st = dict(feat=ops.unpack_nhwc16(self.fw_feat, self.C), flow=self.fw_flow, feat_up=ops.unpack_nhwc16(self.fw_feat_up, self.C),
                  conf=self.fw_conf, frame_itr_num=self.frame_itr_num)
        kf = getattr(self, 'keyframe_idx', None)
        if kf is not None:
            st['keyframe_idx'] = [int(k) for k in kf]
        return st

    def import_state(self, st):
        cs = (self.C + 7) // 8 * 8                              # channel stride of the maps (C = 36 -> 40, zero padding)
        self.fw_feat = ops.pack_nhwc16(st['feat'].contiguous(), cs)
        self.fw_flow = st['flow'].contiguous()
        self.fw_feat_up = ops.pack_nhwc16(st['feat_up'].contiguous(), cs)
        self.fw_conf = st['conf'].contiguous()
        self.frame_itr_num = int(st['frame_itr_num'])
        if st.get('keyframe_idx') is not None and hasattr(self, 'keyframe_idx'):
            import numpy as np
            self.keyframe_idx = np.asarray(st['keyframe_idx'], dtype=np.int64)

    # ---- the same state as ONE contiguous device buffer in the engine's native layouts (the multi-GPU hand-off message):
    #      [64-byte header: int32 frame_itr_num, h, w, Cs (channel STRIDE of the maps: 24 / 48 / 40 for C = 36), number of
    #       key-frame indices, up to 11 key-frame indices (RefVSR_IR)] [feat fp16 HWC] [feat_up fp16 HWC] [flow fp32 planar]
    #      [conf fp32 planar] = (5 Cs * 2 + 12) * h*w + 64 bytes -- half the bytes of the planar fp32 export, no unpack / pack
    #      kernels, one message
    STATE_HEADER = 64

    def _state_cs(self):
        return (self.C + 7) // 8 * 8

    def state_nbytes(self, h, w):
        cs = self._state_cs()
        return self.STATE_HEADER + h * w * (2 * cs + 8 * cs + 8 + 4)

    def export_state_packed(self):
        if self.fw_feat is None:
            return None
        self._await_fw_up(self.fw_feat_up)
        h, w, cs = self.fw_feat.shape
        assert cs == self._state_cs() and self.fw_feat_up.shape[2] == cs
        buf = torch.empty(self.state_nbytes(h, w), dtype=torch.uint8, device=self.fw_feat.device)
        kf = [int(k) for k in (getattr(self, 'keyframe_idx', None) if getattr(self, 'keyframe_idx', None) is not None else [])]
        assert len(kf) <= 11
        hdr = [self.frame_itr_num, h, w, cs, len(kf)] + kf + [0] * (11 - len(kf))
        buf[:self.STATE_HEADER].view(torch.int32).copy_(torch.tensor(hdr, dtype=torch.int32), non_blocking=False)
        o = self.STATE_HEADER
        for t in (self.fw_feat, self.fw_feat_up, self.fw_flow, self.fw_conf):
            n = t.numel() * t.element_size()
            buf[o:o + n].view(t.dtype).copy_(t.reshape(-1))
            o += n
        return buf

    def import_state_packed(self, buf):
        hdr = buf[:self.STATE_HEADER].view(torch.int32).cpu().tolist()
        itr, h, w, cs, nk = hdr[:5]
        assert cs == self._state_cs() and buf.numel() == self.state_nbytes(h, w), 'state buffer does not match this model'
        dev = buf.device
        o = self.STATE_HEADER

        def take(shape, dtype):
            nonlocal o
            n = 1
            for d in shape:
                n *= d
            n *= torch.empty((), dtype=dtype).element_size()
            t = buf[o:o + n].view(dtype).view(shape).clone()
            o += n
            return t
        self.fw_feat = take((h, w, cs), torch.float16)
        self.fw_feat_up = take((2 * h, 2 * w, cs), torch.float16)
        self.fw_flow = take((2, h, w), torch.float32)
        self.fw_conf = take((1, h, w), torch.float32)
        self.frame_itr_num = int(itr)
        if hasattr(self, 'keyframe_idx') and nk > 0:
            import numpy as np
            self.keyframe_idx = np.asarray(hdr[5:5 + nk], dtype=np.int64)
        assert dev == self.fw_feat.device

    # ---- the hand-off in two messages (shard.run_wavefront; VERDICT r3 "predicted >= 5.5 x": the restart-free clip is bound by its
    #      B1 chain): [header | feat | flow | conf] = (2 Cs + 12) h w + 64 bytes, what a forward-branch step touches FIRST, and the 2x
    #      state feat_up (8 Cs h w bytes, 76 % of the state, sent as it lies -- no packing copy), which the step reads only after
    #      its residual chain has been launched (_prop_step): the receiver starts its step when the small message is there and waits
    #      for the large one in the middle of it (_await_fw_up).
    split_state_ok = True                # (EngineIR reads fw_feat_up outside _prop_step: it keeps the one-message form)

    def state_head_nbytes(self, h, w):
        return self.STATE_HEADER + h * w * (2 * self._state_cs() + 8 + 4)

    def export_state_split(self):
        """(head: one uint8 device buffer, feat_up: the 2x state tensor itself)."""
        if self.fw_feat is None:
            return None
        self._await_fw_up(self.fw_feat_up)
        h, w, cs = self.fw_feat.shape
        assert cs == self._state_cs() and not hasattr(self, 'keyframe_idx')
        buf = torch.empty(self.state_head_nbytes(h, w), dtype=torch.uint8, device=self.fw_feat.device)
        hdr = [self.frame_itr_num, h, w, cs, 0] + [0] * 11
        buf[:self.STATE_HEADER].view(torch.int32).copy_(torch.tensor(hdr, dtype=torch.int32), non_blocking=False)
        o = self.STATE_HEADER
        for t in (self.fw_feat, self.fw_flow, self.fw_conf):
            n = t.numel() * t.element_size()
            buf[o:o + n].view(t.dtype).copy_(t.reshape(-1))
            o += n
        return buf, self.fw_feat_up

    def import_state_head(self, buf, feat_up, wait_feat_up):
        """The small message + the tensor the large one is being received into; wait_feat_up(): makes the CURRENT stream wait for
        it (called once, by the first reader of fw_feat_up)."""
        hdr = buf[:self.STATE_HEADER].view(torch.int32).cpu().tolist()
        itr, h, w, cs = hdr[:4]
        assert cs == self._state_cs() and buf.numel() == self.state_head_nbytes(h, w), 'state buffer does not match this model'
        assert tuple(feat_up.shape) == (2 * h, 2 * w, cs) and feat_up.dtype == torch.float16
        o = self.STATE_HEADER

        def take(shape, dtype):
            nonlocal o
            n = torch.empty((), dtype=dtype).element_size()
            for d in shape:
                n *= d
            t = buf[o:o + n].view(dtype).view(shape).clone()
            o += n
            return t
        self.fw_feat = take((h, w, cs), torch.float16)
        self.fw_flow = take((2, h, w), torch.float32)
        self.fw_conf = take((1, h, w), torch.float32)
        self.fw_feat_up = feat_up
        self._fw_up_wait = wait_feat_up
        self.frame_itr_num = int(itr)

    def _await_fw_up(self, feat_up):
        """Before the first read of a 2x state that may still be arriving (split hand-off)."""
        if self._fw_up_wait is not None and feat_up is self.fw_feat_up:
            w, self._fw_up_wait = self._fw_up_wait, None
            w()

    # ---- per-frame contexts ahead of their windows (shard.run_wavefront with exchange_contexts: every context is prepared by ONE
    #      rank and sent to the others whose windows need it).  A context = everything prepare_frame derives from one (lr_i, ref_i)
    #      pair (RefVSR.py:196-204,233-234,127,136); the message carries the four expensive maps [conf fp32 | idx int32 | aligned
    #      fp16 HWC | aligned_up fp16 HWC] as they lie in HBM (32.2 MB for RefVSR_small at 270p), the cheap ones (the 8-channel
    #      copy of the frame, the SPyNet pyramid) are rebuilt by the receiver from its own copy of the frame.
    CTX_FIELDS = ('conf', 'idx', 'aligned', 'aligned_up')

    def _ctx(self, fid):
        fr = self.id_cache.get(fid)
        return fr if fr is not None else self.ctx_pinned.get(fid)

    @torch.no_grad()
    def prepare_context(self, lr, ref, fid):
        """The context of frame `fid` -- its id in THIS engine's forward(frame_ids=...) / phase_a: Network names frame f of batch
        element b (b, f) -- prepared now on the current stream and kept until a window that contains the id has used it."""
        assert self.cache, 'contexts are kept by the id-keyed window cache (config.cache_windows)'
        fr = self._ctx(fid)
        if fr is None:
            fr = self.ctx_pinned[fid] = FrameCtx(lr, ref)
        with torch.cuda.device(lr.device), ops.on_stream(torch.cuda.current_stream(lr.device)):
            self.pyramid(fr)
            strict, self.ctx_strict = self.ctx_strict, False
            try:
                self.prepare_frame(fr)
            finally:
                self.ctx_strict = strict
        return fr

    def context_spec(self, fid):
        """[(field, shape, dtype)] of the message of a PREPARED context: what a receiver needs to size and slice it."""
        fr = self._ctx(fid)
        assert fr is not None and fr.conf is not None, 'context %r is not prepared' % (fid,)
        return [(k, tuple(getattr(fr, k).shape), getattr(fr, k).dtype) for k in self.CTX_FIELDS]

    @staticmethod
    def context_nbytes(spec):
        n = 0
        for _, shape, dtype in spec:
            m = torch.empty((), dtype=dtype).element_size()
            for d in shape:
                m *= d
            n += (m + 15) // 16 * 16
        return n

    @torch.no_grad()
    def export_context(self, fid):
        """One uint8 device buffer holding CTX_FIELDS of a prepared context, 16-byte aligned parts (current stream)."""
        spec = self.context_spec(fid)
        fr = self._ctx(fid)
        buf = torch.empty(self.context_nbytes(spec), dtype=torch.uint8, device=fr.conf.device)
        o = 0
        for k, shape, dtype in spec:
            t = getattr(fr, k)
            n = t.numel() * t.element_size()
            buf[o:o + n].view(dtype).copy_(t.reshape(-1))
            o += (n + 15) // 16 * 16
        return buf

    @torch.no_grad()
    def import_context(self, lr, ref, fid, buf, spec):
        """Install a context another rank prepared: its maps are views into `buf` (kept alive by the context), the frame copy, its
        8-channel HWC form and the SPyNet pyramid are made here (current stream)."""
        assert self.cache and buf.dtype == torch.uint8 and buf.numel() == self.context_nbytes(spec)
        # A window that merely CONTAINS the frame (positions before its centre) leaves an unprepared FrameCtx in the id cache
        # (_frames): e.g. frame_num = 5 and a block starting one frame before a reset frame -- that context is filled in place,
        # like prepare_context does.  Only a context that is already PREPARED is a protocol error.
        fr = self._ctx(fid)
        assert fr is None or fr.conf is None, 'context %r exists already (prepared)' % (fid,)
        pinned = fr is None
        if fr is None:
            fr = FrameCtx(lr, ref)
        with torch.cuda.device(lr.device), ops.on_stream(torch.cuda.current_stream(lr.device)):
            if fr.lr8 is None:
                fr.lr8 = ops.pack_nhwc16(fr.lr, 8)
            self.pyramid(fr)
        o = 0
        for k, shape, dtype in spec:
            n = torch.empty((), dtype=dtype).element_size()
            for d in shape:
                n *= d
            setattr(fr, k, buf[o:o + n].view(dtype).view(shape))
            o += (n + 15) // 16 * 16
        if pinned:
            self.ctx_pinned[fid] = fr
        return fr

    # ------------------------------------------------------------------ building blocks
    def cw(self, name):
        return self.W.conv[name]

    def _zeros(self, shape, dtype, dev):
        """Read-only zero maps (the start of a propagation branch, RefVSR.py:211-214,244-247): allocated and filled once per
        geometry instead of three fill launches (31 MB at 270p) per call.  Never written by any kernel; the one-time host
        synchronisation makes them valid on every stream."""
        key = (tuple(shape), dtype, str(dev))
        z = self._zero_maps.get(key)
        if z is None:
            if len(self._zero_maps) > 12:
                # kernels queued on the internal streams may still be reading the maps that are dropped here (they carry no
                # record_stream): the device is drained first, then the allocator may reuse them (reached only by a
                # process that walks through more than twelve geometries)
                torch.cuda.synchronize(dev)
                self._zero_maps.clear()
            z = self._zero_maps[key] = torch.zeros(shape, dtype=dtype, device=dev)
            torch.cuda.current_stream(dev).synchronize()
        return z

    def _block_chain(self, x, pairs, act):
        """A run of residual blocks x <- x + conv2(act(conv1 x)); pairs = [(conv1, conv2), ...] packed weights.
        One launch per block (fused kernel) or two launches per block (fuse_resblocks off / unsupported channel count).
        (A kernel chaining TWO blocks per launch with halo recomputation was built in round 1 and measured in round 2:
        29.3 us vs 2 x 14.7 us on the LR maps, slower on the 2x maps and 4 % slower end to end -- removed.)"""
        if self.fuse_resblocks and self.rb24 and x.shape[2] == 24 and pairs[0][0].raw is not None:
            # mid_channels = 24 (the RefVSR_small family): the compile-time-specialised kernel, one blob per block
            chains = self.W.chains                                  # lives and dies with the packed weights it points into
            key = ('rb24',) + tuple(id(c1) for c1, _ in pairs)
            ch = chains.get(key)
            if ch is None:
                ch = chains[key] = ops.Resblock24Chain(pairs, x.device)
            if self.chain_events is None:
                return ops.resblock24_chain(ch, x, act)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = ops.resblock24_chain(ch, x, act)
            e1.record()
            self.chain_events.append((e0, e1, len(pairs), x.shape[0], x.shape[1]))
            return out
        if (self.rb48 and x.shape[2] == 48 and pairs[0][0].raw is not None and 0.0 <= act <= 1.0 and
                x.shape[0] * x.shape[1] <= self.rb48_max_pixels):
            # mid_channels = 48 (RefVSR / RefVSR_MFID / RefVSR_MFID_8K): one launch per block, the two 84 KB weight sets swap
            # per tile through LDS-DMA (csrc/resblock48.hip); the intermediate map never reaches HBM.  Maps up to 540 x 960:
            # per block 12.9 vs 23.5 us at 135 x 240, 26.7 vs 30.2 us at 270 x 480 (RefVSR_MFID 85.0 vs 83.2 frames/s); at
            # 1080 x 1920 the two forms are equal stand-alone (396 vs 405 us: 32 tiles per workgroup, each swapping 168 KB of
            # weights, against conv48's 16 x 32 tiles on sixteen waves) and the 1080p -> 8K frame is 0.7 % slower with the fused
            # block (6.76 vs 6.81 frames/s) -> the large maps keep the two launches (profiles/r04_resblock48.txt)
            chains = self.W.chains
            key = ('rb48',) + tuple(id(c1) for c1, _ in pairs)
            ch = chains.get(key)
            if ch is None:
                ch = chains[key] = ops.Resblock48Chain(pairs, x.device)
            return ops.resblock48_chain(ch, x, act)
        if self.fuse_resblocks and self.chain_calls and ops.resblock_chain_ok(x.shape[2]):
            # one library call per run (same launches, same results): 156 of the ~330 launches of a frame
            chains = self.W.chains
            key = tuple(id(c1) for c1, _ in pairs)
            ch = chains.get(key)
            if ch is None:
                ch = chains[key] = ops.ResblockChain(pairs)
            return ops.resblock_chain(ch, x, act)
        for c1, c2 in pairs:
            if self.fuse_resblocks:
                x = ops.resblock(c1, c2, x, act=act)
            else:
                t = ops.conv(c1, x, act=act)
                x = ops.conv(c2, t, res=x)
        return x

    def res_list(self, x, name, n):
        """ResList (RefVSR_/common.py:64-82) with ResBlocks (:25-39)."""
        pairs = [(self.cw('%s.RBs.%d.conv1' % (name, i)), self.cw('%s.RBs.%d.conv2' % (name, i))) for i in range(n)]
        y = self._block_chain(x, pairs, 0.2)
        return ops.conv(self.cw(name + '.conv_tail'), y, res=x)

    def resblocks(self, lr8, feat, name, stop=None, resume=None):
        """ResidualBlocksWithInputConv (RefVSR.py:327-360); torch.cat([lr, feat]) fused as two sources.
        stop = n: only the input conv and the first n blocks (returns the intermediate map); resume = (n, map): the blocks from n
        on -- the two halves of one call, for running them on different streams (same launches, same results)."""
        pairs = [(self.cw('%s.main.2.%d.conv1' % (name, i)), self.cw('%s.main.2.%d.conv2' % (name, i)))
                 for i in range(self.nb)]
        if resume is not None:
            n, x = resume
            return self._block_chain(x, pairs[n:], 0.0) if n < self.nb else x
        x = ops.conv(self.cw(name + '.main.0'), lr8, feat, act=0.1)
        if stop is not None:
            return self._block_chain(x, pairs[:stop], 0.0) if stop > 0 else x
        return self._block_chain(x, pairs, 0.0)

    def pyramid(self, fr):
        """SPyNet.forward resize-to-/32 + normalise + 5x avg_pool2d (SPyNet.py:62-81,117-126)."""
        if fr.pyr is None:
            h, w = fr.lr.shape[1:]
            w_up = w if w % 32 == 0 else 32 * (w // 32 + 1)
            h_up = h if h % 32 == 0 else 32 * (h // 32 + 1)
            lv = [ops.resize(fr.lr, (h_up, w_up), ops.RS_BILINEAR, mean=VGG_MEAN, std=VGG_STD)]
            for _ in range(5):
                lv.append(ops.avgpool2(lv[-1]))
            fr.pyr = lv[::-1]
        return fr.pyr

    def flow(self, fr_ref, fr_supp, share=None):
        """FlowNet(ref, supp) (SPyNet.py:49-139), cached per ordered frame pair.  `share`: streams that will consume
        the result besides the current one (pipelined mode): the tensor is recorded on them and carries an event."""
        key = (fr_ref.uid, fr_supp.uid)
        hit = self.flow_cache.get(key)
        if hit is not None:
            out, ev = hit
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
                out.record_stream(torch.cuda.current_stream())
            return out
        for f in (fr_ref, fr_supp):
            if f.ready is not None:
                torch.cuda.current_stream().wait_event(f.ready)
        pr, ps = self.pyramid(fr_ref), self.pyramid(fr_supp)
        h, w = fr_ref.lr.shape[1:]
        flow = None
        for lvl in range(6):
            x, fup = ops.spynet_level_input(pr[lvl], ps[lvl], flow)
            p = 'FlowNet.basic_module.%d.basic_module.' % lvl
            small = x.shape[0] * x.shape[1] <= self.spynet_mt1_pixels
            for j in range(4):
                cw = self.W.conv.get(p + '%d.conv/mt1' % j) if small else None
                x = ops.conv(cw if cw is not None else self.cw(p + '%d.conv' % j), x, act=0.0)
            flow = ops.conv(self.cw(p + '4.conv'), x, planar_out=True, res_planar=fup)
        h_up, w_up = flow.shape[1:]
        out = ops.resize(flow, (h, w), ops.RS_BILINEAR, chan_mul=[float(w) / float(w_up), float(h) / float(h_up)])
        ev = None
        if share:
            ev = torch.cuda.Event()
            ev.record()
            for st in share:
                out.record_stream(st)
        self.flow_cache[key] = (out, ev)
        return out

    def flows(self, pairs, share=None):
        """FlowNet for several ordered frame pairs [(ref frame, supp frame), ...].  The pairs that are not cached yet are
        computed TOGETHER, one launch per layer over all of them (RefvsrConv.batch; the reference calls SPyNet once per pair,
        SPyNet.py:49-104, RefVSR.py:182-191): same weights, same shapes, and the coarse pyramid levels are launches of 2-36
        workgroups -- two flows per launch cost about what one does.  Results equal flow() pair by pair, bit for bit."""
        todo = []
        for a, b in pairs:
            if (a.uid, b.uid) not in self.flow_cache and all((a.uid, b.uid) != (x.uid, y.uid) for x, y in todo):
                todo.append((a, b))
        if len(todo) >= 2 and self.spynet_batch:
            for i0 in range(0, len(todo), 8):         # (a frame group of four asks for eight flows: ONE pass -- the coarse levels are
                chunk = todo[i0:i0 + 8]               #  launches of 2-36 workgroups per image; 222.5 vs 220.6 frames/s with two passes of four)
                if len(chunk) >= 2:
                    self._flow_batch(chunk, share)
        return [self.flow(a, b, share) for a, b in pairs]

    def _flow_batch(self, todo, share):
        cur = torch.cuda.current_stream()
        for a, b in todo:
            for f in (a, b):
                if f.ready is not None:
                    cur.wait_event(f.ready)
        pr = [self.pyramid(a) for a, _ in todo]
        ps = [self.pyramid(b) for _, b in todo]
        B = len(todo)
        h, w = todo[0][0].lr.shape[1:]
        flow = None
        for lvl in range(6):
            x, fup = ops.spynet_level_input_batch([p_[lvl] for p_ in pr], [p_[lvl] for p_ in ps], flow)
            p = 'FlowNet.basic_module.%d.basic_module.' % lvl
            small = x.shape[1] * x.shape[2] <= self.spynet_mt1_pixels
            for j in range(4):
                cw = self.W.conv.get(p + '%d.conv/mt1' % j) if small else None
                x = ops.conv(cw if cw is not None else self.cw(p + '%d.conv' % j), x, act=0.0, batch=B)
            flow = ops.conv(self.cw(p + '4.conv'), x, planar_out=True, res_planar=fup, batch=B)
        h_up, w_up = flow.shape[2:]
        # back to the frame size, x / y components rescaled (SPyNet.py:128-137): refvsr_resize takes <= 4 channels with
        # per-channel factors = two flows per launch
        cm = [float(w) / float(w_up), float(h) / float(h_up)]
        outs = [ops.resize(flow[b0:b0 + 2].reshape(-1, h_up, w_up), (h, w), ops.RS_BILINEAR, chan_mul=cm * min(2, B - b0))
                for b0 in range(0, B, 2)]
        out = outs[0].view(-1, 2, h, w) if len(outs) == 1 else torch.cat([o_.view(-1, 2, h, w) for o_ in outs], 0)
        ev = None
        if share:
            ev = torch.cuda.Event()
            ev.record()
            for st in share:
                out.record_stream(st)
        for i, (a, b) in enumerate(todo):
            self.flow_cache[(a.uid, b.uid)] = (out[i], ev)

    def feature_match(self, fr):
        """FeatureMatching.forward (RefVSR_/attention.py:58-100), fused GEMM+argmax."""
        R = self.W.raw
        h, w = fr.lr.shape[1:]

        fe = 'feature_match.feature_extract.'

        def extract(x):       # VGG19 head + 1x1 map in exact fp32 on v_mfma_f32_16x16x4_f32 (attention.py:31-42)
            x = ops.pack_nhwc32(x, 4)
            x = ops.conv(self.cw(fe + '0'), x, act=0.0)
            if not self.vgg7:
                x = ops.conv(self.cw(fe + '2'), x, act=0.0)
                if self.map1x1:
                    return ops.conv1x1_f32(x, *R[fe + 'map64.0'], act=0.2)
                return ops.conv(self.cw(fe + 'map64.0'), x, act=0.2, planar_out=True)
            x = ops.conv(self.cw(fe + '2'), x, act=0.0, planar_out=True)
            x = ops.pack_nhwc32(ops.maxpool2(x))                                   # VGG19[4]
            x = ops.conv(self.cw(fe + '5'), x, act=0.0)
            if self.map1x1:
                return ops.conv1x1_f32(x, *R[fe + 'map128.0'], act=0.2)
            return ops.conv(self.cw(fe + 'map128.0'), x, act=0.2, planar_out=True)
        lr_n = ops.conv_direct(fr.lr, *R['feature_match.sub_mean'])
        ref_n = ops.conv_direct(fr.ref, *R['feature_match.sub_mean'])
        if self.hd:                                                                # attention.py:65-67
            f = 1.0 / (self.cfg.scale // 2)
            oh, ow = int(h * f), int(w * f)
            lr_n = ops.resize(lr_n, (oh, ow), ops.RS_NEAREST, (1.0 / f, 1.0 / f))
            ref_n = ops.resize(ref_n, (oh, ow), ops.RS_NEAREST, (1.0 / f, 1.0 / f))
        lr_f = extract(lr_n)
        ref_f = extract(ops.avgpool2(ref_n))
        lr_rows, inv_lr, lr_lo = ops.match_patches(lr_f, ops.hip.MATCH_COLBLOCK, want_lo=True)
        ref_rows, inv_ref, ref_lo = ops.match_patches(ref_f, ops.hip.MATCH_ROWCHUNK, want_lo=True)
        n_lr = lr_f.shape[1] * lr_f.shape[2]
        n_ref = ref_f.shape[1] * ref_f.shape[2]
        if self.kernel_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        cand, cand_val = ops.match_top2(ref_rows, n_ref, lr_rows, n_lr, self.match_row_splits)
        if self.kernel_events is not None:
            e1.record()
            self.kernel_events.append((e0, e1))
        # exact fp32 re-rank of the top-2 + exhaustive fp32-grade (split-fp16) search of the columns the fp16 GEMM cannot decide
        conf, idx, _ = ops.match_refine(lr_f, ref_f, inv_lr, inv_ref, cand, cand_val, self.match_margin, (lr_rows, lr_lo),
                                        (ref_rows, ref_lo))
        conf = conf.view(1, lr_f.shape[1], lr_f.shape[2])
        grid = (lr_f.shape[1], lr_f.shape[2])
        if grid[0] != h:                                                           # attention.py:96-98 (HD)
            conf = ops.bicubic_scale(conf, float(h) / grid[0], clamp01=True)
        return conf, idx, grid

    def aligned_conv(self, feats, query, rgb8, prefix, ks):
        """AlignedConv2d.forward (RefVSR_/alignment.py:39-100); query: planar fp32 [3,.,.] (bicubic x2 inside)."""
        def enc(z8):
            e = ops.conv(self.cw(prefix + '.conv1.0'), z8, act=0.2)
            t = ops.conv(self.cw(prefix + '.conv1.2.conv1'), e, act=0.2)
            return ops.conv(self.cw(prefix + '.conv1.2.conv2'), t, res=e, post=0.2)
        q = enc(ops.bicubic_scale(query, 2, clamp01=False, nhwc16_out=True))
        r = enc(rgb8)
        a = ops.conv(self.cw(prefix + '.p_conv.0'), r, q, stride=ks, act=0.2)
        t = ops.conv(self.cw(prefix + '.p_conv.2.conv1'), a, act=0.2)
        a = ops.conv(self.cw(prefix + '.p_conv.2.conv2'), t, res=a, post=0.2)
        affine = ops.conv(self.cw(prefix + '.p_conv.4'), a, planar_out=True, add_const=1.0, clamp=(-3.0, 3.0))
        return ops.aligned_sample(feats, affine, ks)

    def _ref_encoders(self, fr):
        """ref_feat = res1(ref_encoder1(ref)), ref_feat_down = res2(ref_encoder2(ref_feat))  (RefVSR.py:233-234)."""
        ref8 = ops.pack_nhwc16(fr.ref, 8)
        x = ops.conv(self.cw('ref_encoder1.0.0'), ref8, act=0.2)
        x = ops.conv(self.cw('ref_encoder1.1.0'), x, act=0.2)
        ref_feat = self.res_list(x, 'res1', 4)
        x = ops.conv(self.cw('ref_encoder2.0.0'), ref_feat, stride=2, act=0.2)
        x = ops.conv(self.cw('ref_encoder2.1.0'), x, act=0.2)
        return ref_feat, self.res_list(x, 'res2', 4)

    def prepare_frame(self, fr):
        """Everything that is a function of (lr_i, ref_i) only: matching (RefVSR.py:196-204), reference
        encoders (:233-234) and both AlignedAttention outputs (:127,136)."""
        if fr.conf is not None:
            return
        if self.ctx_strict:
            # shard.run_wavefront(exchange_contexts=True): every context a window needs was prepared ahead (prepare_context) or
            # received (import_context); preparing one here would silently redo another rank's work
            raise RuntimeError('a window needs a per-frame context that was neither prepared ahead nor imported (strict mode)')
        h, w = fr.lr.shape[1:]
        fr.lr8 = ops.pack_nhwc16(fr.lr, 8)
        # (running the reference encoders on a side stream underneath the matching kernel measured neutral, +0..1 %: removed in round 5)
        fr.conf, fr.idx, (gh, gw) = self.feature_match(fr)
        ref_feat, ref_feat_down = self._ref_encoders(fr)
        s1, s2 = self.ks // 2, self.ks
        # aa1 (RefVSR.py:127, attention.py:142-157): gather of LR/2 reference features; with a patch > 1 px (HD)
        # also the affine AlignedConv2d, queried by bicubic x0.5 of the LR frame (RefVSR.py:125)
        feats1 = ops.block_gather_nhwc16(ref_feat_down, fr.idx, gh, gw, s1)
        if s1 > 1:
            rgb1 = ops.block_gather_rgb(fr.ref, fr.idx, gh, gw, s1)
            lr_down = ops.bicubic_scale(fr.lr, 0.5, clamp01=True)
            fr.aligned = self.aligned_conv(feats1, lr_down, rgb1, 'aa1.align', s1)
        else:
            fr.aligned = feats1
        feats2 = ops.block_gather_nhwc16(ref_feat, fr.idx, gh, gw, s2)                   # aa2 (RefVSR.py:136)
        rgb2 = ops.block_gather_rgb(fr.ref, fr.idx, gh, gw, s2)                          # attention.py:152-154
        fr.aligned_up = self.aligned_conv(feats2, fr.lr, rgb2, 'aa2.align', s2)

    def rap(self, fr, conf_prop, feat, feat_up):
        """AA_AF_conf_prop (RefVSR.py:123-149)."""
        R = self.W.raw
        fused = self.fuse_conf and ops.conf_alpha_ok(self.cw('conf_fusion.1.0')) and ops.conf_alpha_ok(self.cw('conf_fusion2.1.0'))
        if fused:
            # cat + conf_fusion.0 + conf_fusion.1 (+ the torch.max of :147) in one launch, likewise bicubic x2 + conf_fusion2
            alpha, conf_next = ops.conf_alpha(conf_prop, fr.conf, 1, *R['conf_fusion.0.0'], self.cw('conf_fusion.1.0'), want_max=True)
        else:
            pair = torch.cat([conf_prop, fr.conf], 0)                                    # [2,h,w] (:130)
            a = ops.conv_direct(pair, *R['conf_fusion.0.0'], act=0.2, nhwc16_out=True)
            alpha = ops.conv(self.cw('conf_fusion.1.0'), a, act=0.2)
        t = ops.conv(self.cw('feat_fusion.0.0'), feat, fr.aligned, act=0.2)
        feat = ops.conv(self.cw('feat_fusion.1.0'), t, act=0.2, mul=alpha, res=feat)     # :131
        feat = self.res_list(feat, 'feat_decoder', 8)
        up1 = ops.conv(self.cw('upsample1.upsample_conv'), feat)                         # :138 (pixel shuffle fused)
        feat_up = ops.conv(self.cw('feat_fusion2_1.0.0'), feat_up, up1, act=0.2)
        if fused:
            alpha2 = ops.conf_alpha(conf_prop, fr.conf, 2, *R['conf_fusion2.0.0'], self.cw('conf_fusion2.1.0'))   # :140-142
        else:
            pair_up = ops.bicubic_scale(pair, 2, clamp01=True)                           # :140-141
            a = ops.conv_direct(pair_up, *R['conf_fusion2.0.0'], act=0.2, nhwc16_out=True)
            alpha2 = ops.conv(self.cw('conf_fusion2.1.0'), a, act=0.2)
        t = ops.conv(self.cw('feat_fusion2.0.0'), feat_up, fr.aligned_up, act=0.2)
        feat_up = ops.conv(self.cw('feat_fusion2.1.0'), t, act=0.2, mul=alpha2, res=feat_up)   # :143
        feat_up = self.res_list(feat_up, 'feat_decoder2', 4)
        if not fused:
            conf_next = ops.max2(conf_prop, fr.conf)                                     # :147
        return feat, feat_up, conf_next

    def compute_up(self, bw_up, fw_up, conf_bw, conf_fw, lr_center):
        """compute_up + base (RefVSR.py:104-119,288) + final clamp (:297)."""
        R = self.W.raw
        fus = ops.conv(self.cw('fusion_UP'), bw_up, fw_up)
        if self.fuse_conf and ops.conf_alpha_ok(self.cw('conf_fusion_BWFW.1.0')):
            alpha = ops.conf_alpha(conf_bw, conf_fw, 2, *R['conf_fusion_BWFW.0.0'], self.cw('conf_fusion_BWFW.1.0'))   # :107-109
        else:
            pair_up = ops.bicubic_scale(torch.cat([conf_bw, conf_fw], 0), 2, clamp01=True)
            a = ops.conv_direct(pair_up, *R['conf_fusion_BWFW.0.0'], act=0.2, nhwc16_out=True)
            alpha = ops.conv(self.cw('conf_fusion_BWFW.1.0'), a, act=0.2)
        t = ops.conv(self.cw('feat_fusion_BWFW.0.0'), bw_up, fw_up, act=0.2)
        out = ops.conv(self.cw('feat_fusion_BWFW.1.0'), t, act=0.2, mul=alpha, res=fus)
        out = self.res_list(out, 'feat_decoder_BWFW', 4)
        if self.cfg.scale == 4:                                               # :114-115
            out = ops.conv(self.cw('upsample2.upsample_conv'), out, act=0.1)  # lrelu commutes with pixel_shuffle
        if self.fuse_tail and self.C == 24 and ops.conv_last_ok(24, out.shape[0], out.shape[1]):
            # conv_hr + conv_last + the bicubic base + the clamps in one launch: the HR map between the two convs stays in LDS
            blob = self.W.chains.get('conv_hr_last_blob')
            if blob is None:
                from .packing import pack_conv_hr_last
                blob = self.W.chains['conv_hr_last_blob'] = pack_conv_hr_last(*self.cw('conv_hr').raw, *self.cw('conv_last').raw).to(out.device).contiguous()
            return ops.conv_hr_last(blob, out, lr_center, act=0.1, result_dtype=self.result_dtype)
        out = ops.conv(self.cw('conv_hr'), out, act=0.1)
        if self.fuse_head and ops.conv_last_ok(self.C, out.shape[0], out.shape[1]):
            # conv_last + the bicubic base + the clamps in one launch: the base map is evaluated per output value
            blob = self.W.chains.get('conv_last_blob')
            if blob is None:
                from .packing import pack_conv_last
                blob = self.W.chains['conv_last_blob'] = pack_conv_last(*self.cw('conv_last').raw).to(out.device).contiguous()
            return ops.conv_last(blob, out, lr_center, result_dtype=self.result_dtype)
        base = ops.bicubic_scale(lr_center, self.cfg.scale, clamp01=True)
        return ops.convert_result(ops.conv(self.cw('conv_last'), out, planar_out=True, res_planar=base, clamp=(0.0, 1.0)), self.result_dtype)

    # ------------------------------------------------------------------ window bookkeeping
    def _frames(self, lrs, refs, frame_ids=None):
        """Wrap the t frames of this window, reusing per-frame contexts of the previous window (or of
        earlier positions in this window) whose content is identical -- or, when the caller supplies frame ids,
        whose id matches (no device work, no synchronisation)."""
        t = lrs.shape[0]
        frames = [None] * t
        if frame_ids is not None and self.cache:
            assert len(frame_ids) == t
            if self.id_cache and next(iter(self.id_cache.values())).lr.shape != lrs.shape[1:]:
                self.id_cache, self.flow_cache = {}, {}
            for i, fid in enumerate(frame_ids):
                fr = self.id_cache.get(fid)
                if fr is None:
                    fr = self.ctx_pinned.pop(fid, None)            # prepared / imported ahead of its first window (prepare_context)
                    if fr is None:
                        fr = FrameCtx(lrs[i], refs[i])
                    self.id_cache[fid] = fr
                frames[i] = fr
            keep = set(frame_ids)
            self.id_cache = {k: v for k, v in self.id_cache.items() if k in keep}
            live = set(f.uid for f in frames)
            self.flow_cache = {k2: v for k2, v in self.flow_cache.items() if k2[0] in live and k2[1] in live}
            self.prev_window = frames
            return frames
        if not self.cache:
            self.flow_cache = {}
            return [FrameCtx(lrs[i], refs[i]) for i in range(t)]
        prev = self.prev_window
        if prev and prev[0].lr.shape != lrs.shape[1:]:          # new clip geometry: nothing to reuse
            prev = []
            self.flow_cache = {}
        if prev:
            # candidates in order of likelihood: the window slid by one (i+1), did not move (i), slid back (i-1);
            # all candidate pairs are compared by ONE kernel launch + one D2H sync
            cand = [(i, i + sh) for sh in (1, 0, -1) for i in range(t) if 0 <= i + sh < len(prev)]
            flags = ops.buffers_equal([x for i, j in cand for x in ((lrs[i], prev[j].lr), (refs[i], prev[j].ref))])
            for n_, (i, j) in enumerate(cand):
                if frames[i] is None and flags[2 * n_] and flags[2 * n_ + 1]:
                    frames[i] = prev[j]
        # repeated frames inside this window (clip edges replicate frames, datasets.py:233-234)
        fresh = [i for i in range(t) if frames[i] is None]
        if len(fresh) > 1:
            pp = [(a, b) for ai, a in enumerate(fresh) for b in fresh[ai + 1:]]
            eq = ops.buffers_equal([x for a, b in pp for x in ((lrs[a], lrs[b]), (refs[a], refs[b]))])
            for n_, (a, b) in enumerate(pp):
                if eq[2 * n_] and eq[2 * n_ + 1]:
                    if frames[a] is None:
                        frames[a] = FrameCtx(lrs[a], refs[a])
                    if frames[b] is None:
                        frames[b] = frames[a]
        for i in range(t):
            if frames[i] is None:
                frames[i] = FrameCtx(lrs[i], refs[i])
        live = set(f.uid for f in frames)
        self.flow_cache = {k2: v for k2, v in self.flow_cache.items() if k2[0] in live and k2[1] in live}
        self.prev_window = frames
        return frames

    # ------------------------------------------------------------------ pipelined forward (opt-in)
    def _pipe_streams(self, dev):
        """Internal streams of the pipelined mode: (M0, M1, F, P).

        Layout (REFVSR_PIPE_LAYOUT / config.pipe_layout), round 4:
          'pf_m' (default)  TWO internal streams: P carries the per-frame preparation AND the forward-branch step (F is P),
                            M the backward branch + upsampler.  With the caller's stream (which only receives the result) that
                            is three streams for the runtime's four hardware queues: which streams share a queue is no longer
                            decided per process.  Round 3 ran P, F and M as three streams (+ the caller's): on boxes where every
                            stream got its own hardware queue THREE kernels ran concurrently -- the 254-workgroup / 152 KB-LDS
                            matching kernel and two conv chains displacing each other CU by CU -- and the rate fell from 187-196 to
                            141-155 frames/s (BENCH_r03: 140.9; profiles/r03_one_vs_two_m_streams.txt).  The fast mode of round 3 was
                            the one in which F and P happened to share a hardware queue, i.e. exactly this layout.
          'pfm'             round 3: P, F, M on three streams (kept for the A/B and the slow-mode reproduction)
          'one'             P, F and M on ONE stream (measurement aid: bench.py times the multi-map launches of a group in it)
        Wider models (C = 48 / 36) additionally alternate two M streams (for mid_channels = 24 two M streams measured equal to one,
        209.5 / 210.2 vs 209.1 / 209.6 frames/s, profiles/r04_two_m_and_mfid_layout_ab.txt: no switch)."""
        # default: P | F | M for the mid_channels = 24 models (what their frame groups run on: an engine whose FIRST pipelined call is a
        # single forward() -- a caller that sends the first window alone -- must not lay its streams out differently and rebuild them at
        # the first group: the idle first set of streams changes the stream -> hardware-queue mapping, 212 vs 231 frames/s,
        # profiles/r05_group_cut_ab.txt); P + F | M for the wider models (round 4's layout, two alternating M streams)
        # The layout is decided ONCE, when the streams are built (re-deriving it per call cost weight look-ups per launch
        # list and could rebuild the streams mid-stream); another layout needs an explicit rebuild: `engine._pipe = None` after a
        # device synchronisation (bench.py's single-stream measurement does that), a new device, or a new engine.
        if self._pipe is not None and self._pipe[0].device == dev:
            return self._pipe
        layout = str(getattr(self.cfg, 'pipe_layout', None) or os.environ.get('REFVSR_PIPE_LAYOUT') or self._layout_default or
                     ('pfm' if (self.C == 24 and self.group_ok()) else 'pf_m'))
        if True:
            if layout not in ('pf_m', 'pfm', 'one'):                 # ('p_fm', F on M's stream, measured slower in rounds 4-5: removed in round 6)
                raise ValueError('REFVSR_PIPE_LAYOUT must be pf_m | pfm | one, got %r' % layout)
            two = self.C != 24
            split = str(getattr(self.cfg, 'cu_split', None) or os.environ.get('REFVSR_CU_SPLIT') or '')
            if split and layout in ('pfm', 'pf_m'):
                # CU partitions (round 6, ABI 13; measurement knob, off by default -- DESIGN 5 "CU partitions"): 'p,f,m' CUs for the
                # three streams as disjoint ranges (multiples of 8 = an equal share of every XCD); f = 0 with layout pf_m
                cp, cf, cm = (int(v) for v in split.split(','))
                p_ = ops.CuStream(0, cp, dev)
                f_ = ops.CuStream(cp, cf, dev) if (layout == 'pfm' and cf > 0) else p_
                m = ops.CuStream(cp + (cf if f_ is not p_ else 0), cm, dev)
                self._pipe = [m, m, f_, p_]
                self.pipe_layout = layout
                self._pipe_calls = 0
                return self._pipe
            m = torch.cuda.Stream(device=dev)
            if layout == 'one':                       # measurement aid (bench.py): every section on ONE internal stream -- HIP events
                two = False                           # around a run of launches then bracket nothing but that run
            m2 = torch.cuda.Stream(device=dev) if two else m
            p_ = m if layout == 'one' else torch.cuda.Stream(device=dev)
            f_ = p_ if layout in ('pf_m', 'one') else torch.cuda.Stream(device=dev)
            self._pipe = [m, m2, f_, p_]
            self.pipe_layout = layout
            self._pipe_calls = 0
        return self._pipe

    def _pipe_inputs(self, tensors, input_ready, streams, caller):
        """When the internal streams may read the inputs of a pipelined call.  input_ready = 'materialised' (the caller asserts the
        tensors are final, e.g. a pre-loaded clip): no wait.  An event / stream: the internal streams wait for it (a producer on a copy
        stream keeps the calls pipelined).  None: wait for the caller's stream as it stands -- always safe, but the caller's stream also
        carries the wait for the previous call's result, so consecutive calls then overlap only inside a call."""
        if input_ready is None:
            input_ready = torch.cuda.Event()
            input_ready.record(caller)
        if not isinstance(input_ready, str):
            for st in streams:
                if isinstance(input_ready, torch.cuda.Stream):
                    st.wait_stream(input_ready)
                else:
                    st.wait_event(input_ready)
        elif input_ready != 'materialised':
            raise ValueError("input_ready must be None, 'materialised', a torch.cuda.Event or a torch.cuda.Stream")
        for x in tensors:
            for st in streams:
                x.record_stream(st)

    @torch.no_grad()
    def _forward_pipelined(self, lrs, refs, is_first_frame, want_vis, frame_ids, input_ready=None):
        """Same computation as _forward_seq, spread over the internal streams so that consecutive calls overlap: P prepares the
        window's new frame and its flows while M is still walking the previous call's backward branch and F its forward-branch step.
        Round 6: a steady call IS a frame group of one window (_forward_group_pipelined: one implementation of the P | F | M schedule
        for one window per call, B windows per call and -- forward_multi -- n samples per call); only the gradio mode and a call without
        a carried state that is not a first frame (an error of the caller, reported by _forward_seq) take the reference order on M."""
        dev = lrs.device
        # A window whose forward branch restarts -- a reset_branch roll-over of a running stream (RefVSR.py:168-170) and, since round 6,
        # a caller's FIRST FRAME -- differs from a steady one only in its forward branch: t // 2 + 1 steps from zeros over the window's
        # first frames instead of one step from the carried state.  It stays on the three streams (`rst` in _forward_group_pipelined): no
        # drain of the calls in flight, no serial pass on M, no refill of the pipeline behind it (a first frame used to cost 11.5 ms of
        # serial launches on M at 270p; a clip's first call now overlaps like any other).
        if not bool(self.cfg.EVAL.is_gradio) and (is_first_frame or self.fw_feat is not None):
            outs, vis = self._forward_group_pipelined([(lrs, refs, frame_ids)], input_ready, want_vis, first=bool(is_first_frame))
            return outs[0], vis
        caller = torch.cuda.current_stream()
        M0, M1, F_, P = self._pipe_streams(dev)
        M, Mo = (M0, M1) if (self._pipe_calls & 1) == 0 else (M1, M0)
        self._pipe_calls += 1
        while len(self._inflight) >= self.pipe_depth:
            self._inflight.popleft().synchronize()
        streams = []
        for st in (M0, M1, F_, P):
            if all(st is not s_ for s_ in streams):
                streams.append(st)
        self._pipe_inputs((lrs, refs), input_ready, streams, caller)
        # restart of the forward branch: run the reference order on M, after everything in flight
        for st in (P, F_, Mo):
            M.wait_stream(st)
        with ops.on_stream(M):
            out, vis = self._forward_seq(lrs, refs, True if (self.max_frame_itr_num is not None and self.frame_itr_num == self.max_frame_itr_num)
                                         else is_first_frame, want_vis, frame_ids)
        for st in (F_, P, Mo):
            st.wait_stream(M)
        done = torch.cuda.Event()
        done.record(M)
        caller.wait_event(done)
        out.record_stream(caller)
        if vis:
            for v in vis.values():
                v.record_stream(caller)
        self._inflight.append(done)
        return out, vis

    # ------------------------------------------------------------------ frame groups: multi-map launches (opt-in extension)
    # The backward branches of CONSECUTIVE output frames are independent chains over identical weights (each window restarts its
    # backward branch from zeros, RefVSR.py:211-238): step j of the B chains of a group of B consecutive windows runs as ONE launch
    # per layer over B maps (refvsr_*_batch, ABI 11) -- at 270 x 480 a launch is one 8 x 32 tile per workgroup, a group of four is
    # four tiles per weight fill and per launch (resblock24: 9.9 -> 7.4 us per map, profiles/r05_multimap_microbench.txt).  Only the
    # forward-branch step (which carries the state from frame to frame) stays one frame per launch.  Results are bit-identical to
    # B forward() calls (tests/test_gpu_e2e.py::test_frame_groups_are_bit_identical).
    def group_ok(self):
        """Frame groups run on the default kernels of the mid_channels = 24 family (every launch of the backward branches multi-map)
        and of the mid_channels = 48 family (multi-map warps; its blocks and convs have no multi-map kernels and run map by map inside
        the group's schedule -- same results)."""
        return bool(((self.C == 24 and self.rb24 and self.fuse_resblocks) or (self.C == 48 and self.rb48)) and self.fuse_conf and self.warp_up2 and
                    ops.CONV24 and self.cache and self.overlap and not bool(self.cfg.EVAL.is_gradio) and
                    ops.conf_alpha_ok(self.cw('conf_fusion.1.0')) and ops.conf_alpha_ok(self.cw('conf_fusion2.1.0')))

    def _block_chain_b(self, xs, pairs, act):
        """_block_chain over B maps (lists in, list out): one multi-map launch per block (the 24- and 48-channel fused blocks; anything
        else map by map)."""
        if (self.rb48 and xs[0].shape[2] == 48 and pairs[0][0].raw is not None and 0.0 <= act <= 1.0 and
                xs[0].shape[0] * xs[0].shape[1] <= self.rb48_max_pixels and self.rb48_multimap):
            chains = self.W.chains
            key = ('rb48',) + tuple(id(c1) for c1, _ in pairs)
            ch = chains.get(key)
            if ch is None:
                ch = chains[key] = ops.Resblock48Chain(pairs, xs[0].device)
            return list(ops.resblock48_chain_b(ch, xs, act, stack=False))
        if not (self.rb24 and xs[0].shape[2] == 24 and pairs[0][0].raw is not None):
            return [self._block_chain(x, pairs, act) for x in xs]
        chains = self.W.chains
        key = ('rb24',) + tuple(id(c1) for c1, _ in pairs)
        ch = chains.get(key)
        if ch is None:
            ch = chains[key] = ops.Resblock24Chain(pairs, xs[0].device)
        if self.chain_events is None:
            return list(ops.resblock24_chain_b(ch, xs, act, stack=False))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = list(ops.resblock24_chain_b(ch, xs, act, stack=False))
        e1.record()
        self.chain_events.append((e0, e1, len(pairs), xs[0].shape[0], xs[0].shape[1], len(xs)))
        return out

    def res_list_b(self, xs, name, n):
        pairs = [(self.cw('%s.RBs.%d.conv1' % (name, i)), self.cw('%s.RBs.%d.conv2' % (name, i))) for i in range(n)]
        ys = self._block_chain_b(xs, pairs, 0.2)
        return list(ops.conv_b(self.cw(name + '.conv_tail'), ys, ress=xs, stack=False))

    def resblocks_b(self, lr8s, feats, name, resume=None):
        pairs = [(self.cw('%s.main.2.%d.conv1' % (name, i)), self.cw('%s.main.2.%d.conv2' % (name, i)))
                 for i in range(self.nb)]
        if resume is not None:
            n, xs = resume
            return self._block_chain_b(xs, pairs[n:], 0.0) if n < self.nb else xs
        xs = list(ops.conv_b(self.cw(name + '.main.0'), lr8s, feats, act=0.1, stack=False))
        return self._block_chain_b(xs, pairs, 0.0)

    def rap_b(self, fs, conf_props, feats, feat_ups):
        """rap() of B independent (frame, carried maps) tuples as multi-map launches (the fused-confidence launch list)."""
        R = self.W.raw
        confs = [f.conf for f in fs]
        alpha, conf_next = ops.conf_alpha_b(conf_props, confs, 1, *R['conf_fusion.0.0'], self.cw('conf_fusion.1.0'), want_max=True, stack=False)
        t = ops.conv_b(self.cw('feat_fusion.0.0'), feats, [f.aligned for f in fs], act=0.2, stack=False)
        feat = list(ops.conv_b(self.cw('feat_fusion.1.0'), list(t), act=0.2, muls=list(alpha), ress=feats, stack=False))     # :131
        feat = self.res_list_b(feat, 'feat_decoder', 8)
        up1 = ops.conv_b(self.cw('upsample1.upsample_conv'), feat, stack=False)                                                # :138
        feat_up = list(ops.conv_b(self.cw('feat_fusion2_1.0.0'), feat_ups, list(up1), act=0.2, stack=False))
        alpha2 = ops.conf_alpha_b(conf_props, confs, 2, *R['conf_fusion2.0.0'], self.cw('conf_fusion2.1.0'), stack=False)      # :140-142
        t = ops.conv_b(self.cw('feat_fusion2.0.0'), feat_up, [f.aligned_up for f in fs], act=0.2, stack=False)
        feat_up = list(ops.conv_b(self.cw('feat_fusion2.1.0'), list(t), act=0.2, muls=list(alpha2), ress=feat_up, stack=False))   # :143
        feat_up = self.res_list_b(feat_up, 'feat_decoder2', 4)
        return feat, feat_up, list(conf_next)

    def _prop_step_b(self, fs, branch, feats, feat_ups, confs, fls):
        """_prop_step for B independent chains (lists of B maps / frames / flows; fls None: the first step of the branches)."""
        if fls is None:
            # (frames that a one-frame-per-call window prepared carry the head of this step -- input conv + bw_head_blocks blocks, run
            #  with their preparation: used when every frame of the group has one; a group's own preparation computes none: the whole
            #  first step is cheaper as multi-map launches here)
            heads = [f.bw_head for f in fs] if branch == 'backward_resblocks' else [None]
            if all(hd is not None and hd[0] == heads[0][0] for hd in heads):
                xs = self.resblocks_b(None, None, branch, resume=(heads[0][0], [hd[1] for hd in heads]))
            else:
                xs = self.resblocks_b([f.lr8 for f in fs], feats, branch)
            return self.rap_b(fs, confs, xs, feat_ups)
        confs = list(ops.warp_planar_b(confs, fls, stack=False))
        xs = self.resblocks_b([f.lr8 for f in fs], list(ops.warp_nhwc16_b(feats, fls, stack=False)), branch)
        return self.rap_b(fs, confs, xs, list(ops.warp_nhwc16_up2_b(feat_ups, fls, stack=False)))

    def _frames_group(self, wins):
        """_frames (id-keyed form) for the B windows of a group: the cache keeps the union of their frames."""
        keep = set()
        for _, _, ids in wins:
            keep.update(ids)
        lr0 = wins[0][0]
        if self.id_cache and next(iter(self.id_cache.values())).lr.shape != lr0.shape[1:]:
            self.id_cache, self.flow_cache = {}, {}
        out = []
        for lrs, refs, ids in wins:
            assert len(ids) == lrs.shape[0]
            frames = []
            for i, fid in enumerate(ids):
                fr = self.id_cache.get(fid)
                if fr is None:
                    fr = self.ctx_pinned.pop(fid, None)
                    if fr is None:
                        fr = FrameCtx(lrs[i], refs[i])
                    self.id_cache[fid] = fr
                frames.append(fr)
            out.append(frames)
        self.id_cache = {k: v for k, v in self.id_cache.items() if k in keep}
        live = set(f.uid for fr in out for f in fr)
        self.flow_cache = {k2: v for k2, v in self.flow_cache.items() if k2[0] in live and k2[1] in live}
        self.prev_window = out[-1]
        return out

    def _itr_after(self, k):
        """frame_itr_num as the k-th window from now will find it (the counter restarts at max_frame_itr_num, RefVSR.py:168-170,292-295)."""
        itr = int(self.frame_itr_num)
        for _ in range(k):
            if self.max_frame_itr_num is not None and itr == self.max_frame_itr_num:
                itr = 0
            itr += 1
        return itr

    @torch.no_grad()
    def forward_group(self, wins, is_first_frame=False, input_ready=None):
        """B consecutive windows of this stream in one call: wins = [(lrs [t,3,h,w], refs, frame_ids)] in stream order, every window
        as forward(lrs, refs, is_first, frame_ids=ids) would get it (is_first_frame applies to wins[0]).  Returns [result planar
        [3,s h,s w]] per window -- bit-identical to the B forward() calls.  Runs of >= 2 windows execute as a group (multi-map launches
        on the internal streams: needs set_pipelined(True); a reset_branch roll-over window and -- round 6 -- a caller's first frame stay
        in their group, only their forward branch is the long one); a stream without a carried state, the gradio mode and engines
        without the group schedule run one forward() each.
        input_ready: as in forward() (None | 'materialised' | event | stream), for all windows of the call."""
        outs = [None] * len(wins)
        i = 0
        if is_first_frame:
            self.id_cache, self.flow_cache = {}, {}               # a new clip: ids of the previous one must not match (as in forward())
        if self.group_ok() and self.pipelined:
            # before the first internal stream exists: an engine driven through forward_group runs P | F | M from its first call on
            # (a rebuild later would leave the first set of streams behind, and with more streams than hardware queues the
            # stream -> queue mapping decides which sections serialise); REFVSR_PIPE_LAYOUT / config.pipe_layout override it
            self._layout_default = 'pfm'
        with torch.cuda.device(wins[0][0].device):
            while i < len(wins):
                # the longest run of steady windows from i on (the iteration counter advances by one per window)
                j = i
                head = bool(i == 0 and is_first_frame)              # (round 6: a first frame is a member of its group like a roll-over)
                if (self.group_ok() and self.pipelined and not bool(self.cfg.EVAL.is_gradio) and (self.fw_feat is not None or head)):
                    # (a roll-over window stays inside the group: only its forward branch is the long one, `rst` in
                    #  _forward_group_pipelined)
                    while j < len(wins) and j - i < ops.hip.MAX_MAPS:
                        j += 1
                if j - i >= 2:
                    res, _ = self._forward_group_pipelined(wins[i:j], input_ready, first=head)
                    outs[i:j] = res
                    i = j
                else:
                    lrs, refs, ids = wins[i]
                    outs[i] = self.forward(lrs, refs, bool(is_first_frame) and i == 0, frame_ids=ids, input_ready=input_ready)[0]
                    i += 1
        return outs

    def _forward_group_pipelined(self, wins, input_ready, want_vis=False, first=False):
        """B >= 1 steady windows of this stream on the internal streams (the ONE implementation of the P | F | M schedule, round 6):
        P prepares the windows' new frames and flows, F walks the B forward-branch steps frame by frame, M the B backward branches --
        multi-map launches for B >= 2, the single-map launch list for B = 1 (one forward() per frame: the first layers of its backward
        branch, a function of the new frame alone, run with the frame's preparation on P: `bw_head_blocks`) -- then the upsamplers.
        first: wins[0] is a caller's first frame (RefVSR.py:257-258,292-295): its forward branch starts from zeros like a roll-over's, the
        iteration counter restarts.  Returns (results, vis of the single window or None)."""
        B = len(wins)
        t, _, h, w = wins[0][0].shape
        ctr, dev = t // 2, wins[0][0].device
        for lrs, refs, _ in wins:
            self._check_window(lrs, refs)
            assert tuple(lrs.shape) == (t, 3, h, w)
        if first:
            # the counter as the roll-over logic wants it: the first window restarts (RefVSR.py:292-295: frame_itr_num = 0, then + 1)
            self.frame_itr_num = self.max_frame_itr_num if self.max_frame_itr_num is not None else 0
        else:
            assert tuple(self.fw_feat.shape[:2]) == (h, w), 'frame size changed without is_first_frame=True'
        # windows of the group at which the forward branch restarts (reset_branch roll-over, a caller's first frame): their backward
        # branches are chains like the others', their forward branch is the long one (from zeros over the window's first frames)
        rst = [bool(self.max_frame_itr_num is not None and self._itr_after(b) == self.max_frame_itr_num) for b in range(B)]
        if first:
            rst[0] = True
        caller = torch.cuda.current_stream()
        M0, M1, F_, P = self._pipe_streams(dev)
        # B = 1: the backward branch + upsampler of consecutive calls are independent of each other (only the forward branch carries
        # state): calls alternate between the two M streams of the wider models (mid_channels = 24: M0 is M1, see _pipe_streams).
        # Groups stay on one (two M streams alternating between groups: 210.5 vs 220.4 frames/s, profiles/r05_group_knobs_ab.txt)
        M = M0 if (B > 1 or (self._pipe_calls & 1) == 0) else M1
        self._pipe_calls += 1
        self._await_fw_up(self.fw_feat_up)
        # the host may run at most pipe_depth calls ahead of the GPU (each call in flight holds its intermediates: ~0.3 GB at 270p; a
        # group holds B calls' worth).  Depth 3 since round 4: with 2 the host had ~2.4 ms of slack when it issued a call and a 6 ms
        # hiccup of the host reached the GPU as a gap (profiles/r04_stream_layout_ab.txt)
        while len(self._inflight) >= (self.pipe_depth if B == 1 else max(1, self.pipe_depth - 1)):
            self._inflight.popleft().synchronize()
        streams = []
        for st in (M0, M1, F_, P):
            if all(st is not s_ for s_ in streams):
                streams.append(st)
        self._pipe_inputs([x for lrs, refs, _ in wins for x in (lrs, refs)], input_ready, streams, caller)
        sev = self.stream_events                     # bench.py: per-call (start, end) HIP events of the P / F / M sections
        mark = (lambda st: None) if sev is None else _timed_event
        tev = {'n': B}
        share = (M0, M1, F_, P)

        def publish(f):
            for x in [f.lr, f.ref, f.lr8, f.conf, f.idx, f.aligned, f.aligned_up] + list(f.pyr):
                for st in share:
                    x.record_stream(st)
        # ---- P: everything that is a function of single frames / frame pairs, for all windows of the call
        with ops.on_stream(P):
            tev['P0'] = mark(P)
            n_ctx = next(_uid)
            frs = self._frames_group(wins)                        # new frames are cloned here, on P
            for fr in frs:
                for f in fr:
                    if f.uid > n_ctx:
                        for st in share:
                            f.lr.record_stream(st)
                            f.ref.record_stream(st)
            for b, fr in enumerate(frs):
                for i in range(0 if rst[b] else ctr, t):          # (a roll-over window's forward branch walks its first frames too: cached)
                    f = fr[i]
                    if f.conf is None:
                        self.pyramid(f)
                        self.prepare_frame(f)
                        if B == 1 and i == t - 1 and self.bw_head_blocks >= 0:
                            # (a group's own preparation computes no backward head: its whole first step is cheaper as multi-map launches)
                            zf = self._zeros((h, w, self._state_cs()), torch.float16, dev)
                            f.bw_head = (self.bw_head_blocks, self.resblocks(f.lr8, zf, 'backward_resblocks', stop=self.bw_head_blocks))
                            for st in share:
                                f.bw_head[1].record_stream(st)
                        publish(f)
                        f.ready = torch.cuda.Event()
                        f.ready.record()
                    else:
                        if f.pyr is None:
                            self.pyramid(f)
                        if f.ready is None:          # prepared on M by a first-frame call: make it safe on the other streams too
                            publish(f)
            # the call's new flows in batched SPyNet passes: the backward flows and the forward flow of every window
            need = []
            for b, fr in enumerate(frs):
                need += [(fr[i], fr[i + 1]) for i in range(ctr, t - 1)] + [(fr[ctr + 1], fr[ctr])]
                if rst[b]:
                    need += [(fr[i], fr[i - 1]) for i in range(1, ctr + 1)]           # forward flows of the first frames (cached pairs cost nothing)
            held = self.flows(need, share)                        # (kept until the end of the issue: the cache may trim them)
            tev['P1'] = mark(P)
        # ---- F: the forward-branch steps, one frame after the other (the carried state)
        fws, ev_fw = [], []
        with ops.on_stream(F_):
            tev['F0'] = mark(F_)
            for b, fr in enumerate(frs):
                for f in (fr[:ctr + 2] if rst[b] else (fr[ctr], fr[ctr + 1])):
                    if f.ready is not None:
                        F_.wait_event(f.ready)
                for x in (self.fw_feat, self.fw_feat_up, self.fw_conf, self.fw_flow):
                    if x is not None:                             # (None: a first frame without a previous clip)
                        x.record_stream(F_)
                fw = self._forward_branch(fr, (lambda a, b_, fr=fr: self.flow(fr[a], fr[b_], share)), t, h, w, rst[b])
                for x in fw:
                    x.record_stream(M0)
                    x.record_stream(M1)
                e = torch.cuda.Event(enable_timing=sev is not None)
                e.record()
                fws.append(fw)
                ev_fw.append(e)
            tev['F1'] = ev_fw[-1]
        # ---- M: the B backward branches step by step (B >= 2: multi-map launches), then the upsamplers
        outs, vis = [], None
        with ops.on_stream(M):
            tev['M0'] = mark(M)
            cs_ = self._state_cs()
            feats = [self._zeros((h, w, cs_), torch.float16, dev)] * B
            feat_ups = [self._zeros((2 * h, 2 * w, cs_), torch.float16, dev)] * B
            confs = [self._zeros((1, h, w), torch.float32, dev)] * B
            for i in range(t - 1, ctr - 1, -1):
                fs = [fr[i] for fr in frs]
                for f in fs:
                    if f.ready is not None:
                        M.wait_event(f.ready)
                fls = None
                if i < t - 1:
                    fls = [self.flow(fr[i], fr[i + 1], share) for fr in frs]         # cached by P above: waits on its event
                if B == 1:
                    f1, u1, c1 = self._prop_step(fs[0], 'backward_resblocks', feats[0], feat_ups[0], confs[0], None if fls is None else fls[0])
                    feats, feat_ups, confs = [f1], [u1], [c1]
                else:
                    feats, feat_ups, confs = self._prop_step_b(fs, 'backward_resblocks', feats, feat_ups, confs, fls)
            for b, fr in enumerate(frs):
                M.wait_event(ev_fw[b])
                outs.append(self.compute_up(feat_ups[b], fws[b][1], confs[b], fws[b][2], fr[ctr].lr))
            if want_vis and B == 1:
                vis = collections.OrderedDict()
                vis['conf_map'] = frs[0][ctr].conf
                vis['conf_map_prop'] = ops.max2(confs[0], fws[0][2])
                vis['conf_map_prop_backward'] = confs[0]
                vis['conf_map_prop_forward'] = fws[0][2]
        del held
        self.frame_itr_num = self._itr_after(B)                       # (+ B, through the roll-overs inside the group; RefVSR.py:292-295)
        done = torch.cuda.Event(enable_timing=sev is not None)
        done.record(M)
        if sev is not None:
            tev['M1'] = done
            sev.append(tev)
        caller.wait_event(done)
        for o in outs:
            o.record_stream(caller)
        if vis:
            for v in vis.values():
                v.record_stream(caller)
        self._inflight.append(done)
        return outs, vis

    # ------------------------------------------------------------------ n > 1: the samples of a batch as multi-map launches (round 5)
    # The n samples of a call (lrs [n,t,3,h,w], RefVSR.py:151) are n independent streams over identical weights: in steady state the
    # forward-branch step AND the backward branches of all samples run as multi-map launches (one launch per layer over n maps); only the
    # per-frame preparation stays per sample.  Same streams, events and hazards as a frame group; results equal one forward() per sample,
    # bit for bit (tests/test_gpu_e2e.py::test_batch_samples_as_multimap_launches).
    @staticmethod
    @torch.no_grad()
    def forward_multi(engines, lrs, refs, is_first_frame, frame_ids, input_ready=None):
        """engines: one Engine per sample (sharing one Weights); lrs, refs [n,t,3,h,w]; frame_ids: per-sample id lists.  Returns the n
        results.  Calls that restart a forward branch (first frame, reset_branch roll-over -- the samples' counters run in lock-step),
        more than REFVSR_MAX_MAPS samples and engines without the multi-map launch list run one forward() per sample."""
        n = len(engines)
        lead = engines[0]
        if lead.group_ok() and lead.pipelined:
            for e in engines:                       # P | F | M from the first stream on, like an engine driven through forward_group
                e._layout_default = 'pfm'
        steady = (not is_first_frame and 2 <= n <= ops.hip.MAX_MAPS and lead.group_ok() and
                  all(e.pipelined and e.fw_feat is not None and e.W is lead.W and
                      (e.max_frame_itr_num is None or e.frame_itr_num != e.max_frame_itr_num) and
                      tuple(e.fw_feat.shape[:2]) == tuple(lrs.shape[3:]) for e in engines))
        for e in engines[1:]:                       # one set of internal streams for all samples: the lead's
            e._layout_default, e._pipe, e._inflight = lead._layout_default, lead._pipe, lead._inflight
            if lead._pipe is not None:
                e.pipe_layout, e._pipe_calls = lead.pipe_layout, lead._pipe_calls
        if not steady:
            outs = []
            for b, e in enumerate(engines):
                outs.append(e.forward(lrs[b], refs[b], bool(is_first_frame), frame_ids=frame_ids[b], input_ready=input_ready)[0])
                for e2 in engines:                  # (the lead may just have created the streams)
                    if e2 is not e:
                        e2._pipe, e2._inflight = e._pipe, e._inflight
                        if e._pipe is not None:
                            e2.pipe_layout, e2._pipe_calls = e.pipe_layout, e._pipe_calls
            return outs
        with torch.cuda.device(lrs.device):
            return lead._forward_multi_pipelined(engines, lrs, refs, frame_ids, input_ready)

    def _forward_multi_pipelined(self, engines, lrs, refs, frame_ids, input_ready):
        n, t, _, h, w = lrs.shape
        ctr, dev = t // 2, lrs.device
        for b in range(n):
            self._check_window(lrs[b], refs[b])
        caller = torch.cuda.current_stream()
        M0, M1, F_, P = self._pipe_streams(dev)
        M = M0
        self._pipe_calls += 1
        for e in engines:
            e._pipe, e.pipe_layout, e._pipe_calls = self._pipe, self.pipe_layout, self._pipe_calls
            e._await_fw_up(e.fw_feat_up)
        while len(self._inflight) >= self.pipe_depth:
            self._inflight.popleft().synchronize()
        streams = []
        for st in (M0, M1, F_, P):
            if all(st is not s_ for s_ in streams):
                streams.append(st)
        if input_ready is None:
            input_ready = torch.cuda.Event()
            input_ready.record(caller)
        if not isinstance(input_ready, str):
            for st in streams:
                if isinstance(input_ready, torch.cuda.Stream):
                    st.wait_stream(input_ready)
                else:
                    st.wait_event(input_ready)
        elif input_ready != 'materialised':
            raise ValueError("input_ready must be None, 'materialised', a torch.cuda.Event or a torch.cuda.Stream")
        for st in streams:
            lrs.record_stream(st)
            refs.record_stream(st)
        share = (M0, M1, F_, P)

        def publish(f):
            for x in [f.lr, f.ref, f.lr8, f.conf, f.idx, f.aligned, f.aligned_up] + list(f.pyr):
                for st in share:
                    x.record_stream(st)
        # ---- P: per-sample preparation of the new frames and flows
        frs = []
        with ops.on_stream(P):
            for b, e in enumerate(engines):
                n_ctx = next(_uid)
                fr = e._frames(lrs[b], refs[b], frame_ids[b])
                frs.append(fr)
                for f in fr:
                    if f.uid > n_ctx:
                        for st in share:
                            f.lr.record_stream(st)
                            f.ref.record_stream(st)
                for i in range(ctr, t):
                    f = fr[i]
                    if f.conf is None:
                        e.pyramid(f)
                        e.prepare_frame(f)
                        publish(f)
                        f.ready = torch.cuda.Event()
                        f.ready.record()
                    else:
                        if f.pyr is None:
                            e.pyramid(f)
                        if f.ready is None:
                            publish(f)
                e.flows([(fr[i], fr[i + 1]) for i in range(ctr, t - 1)] + [(fr[ctr + 1], fr[ctr])], share)
        # ---- F: the forward-branch step of all samples as multi-map launches
        with ops.on_stream(F_):
            for e, fr in zip(engines, frs):
                for f in (fr[ctr], fr[ctr + 1]):
                    if f.ready is not None:
                        F_.wait_event(f.ready)
                for x in (e.fw_feat, e.fw_feat_up, e.fw_conf, e.fw_flow):
                    x.record_stream(F_)
            fw_feat, fw_up, fw_conf = self._prop_step_b([fr[ctr] for fr in frs], 'forward_resblocks', [e.fw_feat for e in engines],
                                                        [e.fw_feat_up for e in engines], [e.fw_conf for e in engines],
                                                        [e.fw_flow for e in engines])
            for b, (e, fr) in enumerate(zip(engines, frs)):                               # RefVSR.py:279-283
                e.fw_feat, e.fw_feat_up, e.fw_conf = fw_feat[b], fw_up[b], fw_conf[b]
                e.fw_flow = e.flow(fr[ctr + 1], fr[ctr], share)
            for x in list(fw_up) + list(fw_conf):
                x.record_stream(M0)
                x.record_stream(M1)
            ev_fw = torch.cuda.Event()
            ev_fw.record()
        # ---- M: the backward branches of all samples step by step as multi-map launches, then the upsamplers
        outs = []
        with ops.on_stream(M):
            cs_ = self._state_cs()
            feats = [self._zeros((h, w, cs_), torch.float16, dev)] * n
            feat_ups = [self._zeros((2 * h, 2 * w, cs_), torch.float16, dev)] * n
            confs = [self._zeros((1, h, w), torch.float32, dev)] * n
            for i in range(t - 1, ctr - 1, -1):
                fs = [fr[i] for fr in frs]
                for f in fs:
                    if f.ready is not None:
                        M.wait_event(f.ready)
                fls = None
                if i < t - 1:
                    fls = [e.flow(fr[i], fr[i + 1], share) for e, fr in zip(engines, frs)]
                feats, feat_ups, confs = self._prop_step_b(fs, 'backward_resblocks', feats, feat_ups, confs, fls)
            M.wait_event(ev_fw)
            for b, fr in enumerate(frs):
                outs.append(self.compute_up(feat_ups[b], fw_up[b], confs[b], fw_conf[b], fr[ctr].lr))
        for e in engines:
            e.frame_itr_num += 1
        done = torch.cuda.Event()
        done.record(M)
        caller.wait_event(done)
        for o in outs:
            o.record_stream(caller)
        self._inflight.append(done)
        return outs

    def _side_stream(self, dev):
        """The one side stream of the sequential path (the forward-branch step under the new frame's preparation)."""
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _prop_step(self, f, branch, feat, feat_up, conf, fl, up_from_lr=False):
        """One propagation step (RefVSR.py:216-230 backward, :251-277 forward): warp the carried maps with `fl` (None: first
        step of a branch, nothing to warp), ResidualBlocksWithInputConv, AA_AF_conf_prop.  up_from_lr: the 2x state is warp(warp(feat, fl), flow_up2(fl)) -- the reference's :254 quirk."""
        if fl is None:
            if branch == 'backward_resblocks' and f.bw_head is not None:
                feat = self.resblocks(f.lr8, feat, branch, resume=f.bw_head)      # its head ran with the frame's preparation
            else:
                feat = self.resblocks(f.lr8, feat, branch)
            return self.rap(f, conf, feat, feat_up)
        conf = ops.warp_planar(conf, fl)
        # the 2x state is warped by flow_up2(fl): evaluated inside the warp kernel (no 2x flow map) unless REFVSR_NO_WARP_UP2=1
        warp2 = ops.warp_nhwc16_up2 if self.warp_up2 else (lambda m, fl_: ops.warp_nhwc16(m, ops.flow_up2(fl_)))
        if up_from_lr:
            feat = ops.warp_nhwc16(feat, fl)              # the reference's :254 quirk: the ALREADY WARPED LR state, on the 2x grid
            x = self.resblocks(f.lr8, feat, branch)
            return self.rap(f, conf, x, warp2(feat, fl))
        x = self.resblocks(f.lr8, ops.warp_nhwc16(feat, fl), branch)
        self._await_fw_up(feat_up)                        # (split hand-off: the 2x state may still be arriving -- first read here)
        return self.rap(f, conf, x, warp2(feat_up, fl))

    def _backward_branch(self, fr, flow, t, h, w):
        """Backward propagation branch (RefVSR.py:211-238): restarts from zeros in every window."""
        ctr, dev, cs_ = t // 2, fr[0].lr.device, self._state_cs()
        feat = self._zeros((h, w, cs_), torch.float16, dev)
        feat_up = self._zeros((2 * h, 2 * w, cs_), torch.float16, dev)
        conf = self._zeros((1, h, w), torch.float32, dev)
        for i in range(t - 1, ctr - 1, -1):
            fl = flow(i, i + 1) if i < t - 1 else None    # backward_flows[:, i] = FlowNet(lrs[i], lrs[i+1])
            feat, feat_up, conf = self._prop_step(fr[i], 'backward_resblocks', feat, feat_up, conf, fl)
        return feat_up, conf

    def _forward_branch(self, fr, flow, t, h, w, is_first_frame):
        """Forward propagation branch (RefVSR.py:240-283); updates the carried state.  Runs on the current stream."""
        ctr, dev, cs_ = t // 2, fr[0].lr.device, self._state_cs()
        if is_first_frame:
            feat = self._zeros((h, w, cs_), torch.float16, dev)
            feat_up = self._zeros((2 * h, 2 * w, cs_), torch.float16, dev)
            conf = self._zeros((1, h, w), torch.float32, dev)
            range_start = 0
        else:
            range_start = ctr
        for i in range(range_start, ctr + 1):
            if i > range_start:
                # forward_flows[:, i-1] = FlowNet(lrs[i], lrs[i-1]); :253-255 warps the ALREADY WARPED LR state onto the 2x grid
                feat, feat_up, conf = self._prop_step(fr[i], 'forward_resblocks', feat, None, conf, flow(i, i - 1), up_from_lr=True)
            elif not is_first_frame:                                                # :257-260
                feat, feat_up, conf = self._prop_step(fr[i], 'forward_resblocks', self.fw_feat, self.fw_feat_up, self.fw_conf, self.fw_flow)
            else:
                feat, feat_up, conf = self._prop_step(fr[i], 'forward_resblocks', feat, feat_up, conf, None)
            if i == ctr:                                                            # :279-283
                self.fw_feat, self.fw_feat_up, self.fw_conf = feat, feat_up, conf
                self.fw_flow = flow(ctr + 1, ctr)         # forward_flows[:, ctr]
        return feat, feat_up, conf

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, lrs, refs, is_first_frame, want_vis=False, frame_ids=None, want_log=False, input_ready=None):
        """lrs, refs: cuda float32 [t,3,h,w].  Returns (result planar [3,4h,4w], vis dict or None).
        frame_ids (optional): one hashable id per window frame -> the window cache is keyed by id instead of by
        content comparison; together with set_pipelined(True) it enables cross-call stream pipelining."""
        if is_first_frame and frame_ids is not None:
            # the caller starts a new clip: ids of the previous clip must not match (the internal reset_branch restart
            # of a running clip keeps the cache -- same clip, same ids)
            self.id_cache, self.flow_cache = {}, {}
        self._await_fw_up(self.fw_feat_up)           # (a state imported in two messages whose second part nobody has read yet)
        with torch.cuda.device(lrs.device):          # launches go to the tensors' device, whatever the current device is
            if self.takes_pipelined_path(frame_ids, want_log):
                return self._forward_pipelined(lrs, refs, is_first_frame, want_vis, frame_ids, input_ready)
            cur = torch.cuda.current_stream()
            if isinstance(input_ready, torch.cuda.Stream):
                cur.wait_stream(input_ready)
            elif isinstance(input_ready, torch.cuda.Event):
                cur.wait_event(input_ready)
            with ops.on_stream(cur):
                return self._forward_seq(lrs, refs, is_first_frame, want_vis, frame_ids, want_log)

    def takes_pipelined_path(self, frame_ids, want_log=False):
        """True when forward() with these arguments runs on the internal streams (the only path that reads the inputs off the
        caller's stream order)."""
        return bool(self.pipelined and frame_ids is not None and self.cache and self.overlap and not want_log)

    # ------------------------------------------------------------------ two-phase forward (multi-GPU wavefront)
    def _check_window(self, lrs, refs):
        assert lrs.is_cuda and lrs.dtype == torch.float32 and lrs.dim() == 4 and lrs.shape == refs.shape
        t, _, h, w = lrs.shape
        assert t >= 3 and t % 2 == 1 and h % 2 == 0 and w % 2 == 0, 'need odd t >= 3 and even h, w'
        assert not self.hd or (h % 8 == 0 and w % 8 == 0), 'flag_HD_in needs h, w divisible by 8'
        assert not self.vgg7 or (h % 4 == 0 and w % 4 == 0), 'x2 / HD matching needs h, w divisible by 4'
        return t, h, w

    @torch.no_grad()
    def phase_a(self, lrs, refs, frame_ids=None, first_hint=False):
        """Everything of one forward() that does NOT depend on the carried forward-branch state: frame preparation
        (flows, matching, reference encoders, alignment -- cached across windows as usual) and the whole backward
        branch (RefVSR.py:182-238), i.e. 85-90 % of the work.  Returns a handle for phase_b.  Used by
        shard.run_wavefront: every rank runs phase A of all its frames concurrently, then the ranks run phase B one
        after the other along the state hand-off chain.  first_hint: the call is expected to be a first frame / reset
        frame, prepare frames 0..ctr-1 now instead of lazily in phase B."""
        t, h, w = self._check_window(lrs, refs)
        ctr, dev = t // 2, lrs.device
        with torch.cuda.device(dev), ops.on_stream(torch.cuda.current_stream(dev)):
            zero_flow = torch.zeros((2, h, w), dtype=torch.float32, device=dev) if bool(self.cfg.EVAL.is_gradio) else None
            fr = self._frames(lrs, refs, frame_ids)
            flow = (lambda a, b: zero_flow) if zero_flow is not None else (lambda a, b: self.flow(fr[a], fr[b]))
            if zero_flow is None:
                self.flows([(fr[i], fr[i + 1]) for i in range(ctr, t - 1)] + [(fr[ctr + 1], fr[ctr])] +
                           ([(fr[i], fr[i - 1]) for i in range(1, ctr + 1)] if first_hint else []))
            for i in range(0 if first_hint else ctr, t):
                self.prepare_frame(fr[i])
            bw_up, conf_bw = self._backward_branch(fr, flow, t, h, w)
            # the flows the forward branch will ask for, computed here (parallel phase) and pinned in the handle: the
            # flow cache only keeps pairs of the current window
            flows = {(ctr + 1, ctr): flow(ctr + 1, ctr)}
            if first_hint:
                for i in range(1, ctr + 1):
                    flows[(i, i - 1)] = flow(i, i - 1)
        return dict(fr=fr, flows=flows, zero_flow=zero_flow, bw_up=bw_up, conf_bw=conf_bw, t=t, h=h, w=w)

    @torch.no_grad()
    def phase_a_group(self, wins, first_hints=None, streams=None):
        """phase_a of B <= REFVSR_MAX_MAPS windows of one clip in ONE pass (round 6): wins = [(lrs [t,3,h,w], refs, frame_ids)] -- any
        windows of the clip, consecutive or not (a rank's frames of a block-cyclic partition are not): their backward branches are B
        independent chains over identical weights (RefVSR.py:211-238 restarts from zeros in every window) and run as multi-map launches
        (_prop_step_b: one launch per layer over B maps, the launch list of a frame group's M section), every flow the B windows ask for
        comes out of batched SPyNet passes.  Returns the B handles phase_a would return, bit for bit
        (tests/test_gpu_e2e.py::test_phase_a_group_equals_phase_a); engines without the multi-map launch list run phase_a per window.
        streams=None: everything on the current stream, like phase_a.
        streams=(M, consumers): the stream layout of a frame group for the sharded executor -- the per-frame preparation and
        the flows on the CURRENT stream (the executor's lane a = P), the backward chains on M behind per-frame / per-flow events, so that
        the next group's preparation runs under this group's chains; every handle carries `ready`, an event on M after which all of
        the handle is final; `consumers`: the other streams that will read the contexts and flows (the B1 lane), for the allocator."""
        B = len(wins)
        hints = [False] * B if first_hints is None else [bool(v) for v in first_hints]
        if B > ops.hip.MAX_MAPS or not self.group_ok() or (B == 1 and streams is None):
            return [self.phase_a(lrs, refs, ids, hints[b]) for b, (lrs, refs, ids) in enumerate(wins)]
        t, h, w = self._check_window(wins[0][0], wins[0][1])
        for lrs, refs, ids in wins:
            assert self._check_window(lrs, refs) == (t, h, w) and ids is not None, 'a group needs windows of one geometry, named by frame ids'
        ctr, dev = t // 2, wins[0][0].device
        P = torch.cuda.current_stream(dev)
        M, share = P, None
        if streams is not None:
            M = streams[0]
            share = [P, M] + [st for st in streams[1] if st is not P and st is not M]

        def publish(f):
            for x in [f.lr, f.ref, f.lr8, f.conf, f.idx, f.aligned, f.aligned_up] + list(f.pyr):
                for st in share:
                    x.record_stream(st)
        with torch.cuda.device(dev), ops.on_stream(P):
            frs = self._frames_group(wins)
            need = []
            for b, fr in enumerate(frs):
                need += [(fr[i], fr[i + 1]) for i in range(ctr, t - 1)] + [(fr[ctr + 1], fr[ctr])]
                if hints[b]:
                    need += [(fr[i], fr[i - 1]) for i in range(1, ctr + 1)]
            for b, fr in enumerate(frs):
                for i in range(0 if hints[b] else ctr, t):
                    self.prepare_frame(fr[i])
                    if share is not None and fr[i].ready is None:
                        # (also contexts prepared / imported ahead on this stream: safe on the other streams from here on)
                        if fr[i].pyr is None:
                            self.pyramid(fr[i])
                        publish(fr[i])
                        fr[i].ready = torch.cuda.Event()
                        fr[i].ready.record()
            fl_all = self.flows(need, share)
            if share is not None:
                for fr in frs:                                     # frames a window merely contains: their copies are read by nobody else
                    for f in fr:
                        for x in (f.lr, f.ref):
                            for st in share:
                                x.record_stream(st)
                done_p = torch.cuda.Event()
                done_p.record()
        with torch.cuda.device(dev), ops.on_stream(M):
            if share is not None:
                M.wait_event(done_p)                               # (everything this group reads was produced on P before this point)
                for fl in fl_all:
                    fl.record_stream(M)
            cs_ = self._state_cs()
            feats = [self._zeros((h, w, cs_), torch.float16, dev)] * B
            feat_ups = [self._zeros((2 * h, 2 * w, cs_), torch.float16, dev)] * B
            confs = [self._zeros((1, h, w), torch.float32, dev)] * B
            if B == 1:
                fr = frs[0]
                bw_up, conf_bw = self._backward_branch(fr, (lambda a, b_: self.flow(fr[a], fr[b_], share)), t, h, w)
                feat_ups, confs = [bw_up], [conf_bw]
            else:
                for i in range(t - 1, ctr - 1, -1):
                    fls = None if i == t - 1 else [self.flow(fr[i], fr[i + 1], share) for fr in frs]
                    feats, feat_ups, confs = self._prop_step_b([fr[i] for fr in frs], 'backward_resblocks', feats, feat_ups, confs, fls)
            ready = None
            if share is not None:
                ready = torch.cuda.Event()
                ready.record()
            out = []
            for b, fr in enumerate(frs):
                flows = {(ctr + 1, ctr): self.flow(fr[ctr + 1], fr[ctr], share)}
                if hints[b]:
                    for i in range(1, ctr + 1):
                        flows[(i, i - 1)] = self.flow(fr[i], fr[i - 1], share)
                out.append(dict(fr=fr, flows=flows, zero_flow=None, bw_up=feat_ups[b], conf_bw=confs[b], t=t, h=h, w=w, ready=ready,
                                bw_flows=fl_all if share is not None else None))     # (alive until the executor's final synchronisation)
        return out

    @torch.no_grad()
    def phase_b1(self, pa, is_first_frame):
        """The state-dependent SERIAL part of forward(): the forward-branch step (RefVSR.py:240-283) and the iteration
        counter (:292-295).  Must be called in frame order; leaves the step's outputs in the handle for phase_b2.  The
        multi-GPU wavefront runs B1 of all local frames, hands the state over, and only then runs the upsamplers (B2), so
        that the serial chain over the ranks carries nothing but the forward-branch steps."""
        fr, t, h, w = pa['fr'], pa['t'], pa['h'], pa['w']
        ctr = t // 2
        if self.max_frame_itr_num is not None and self.frame_itr_num == self.max_frame_itr_num:
            is_first_frame = True                                                   # :168-170
        if not is_first_frame and self.fw_feat is None:
            raise RuntimeError('is_first_frame=False but no forward state is held (first call of a stream '
                               'must pass is_first_frame=True, cf. RefVSR.py:257-258)')
        if not is_first_frame and tuple(self.fw_feat.shape[:2]) != (h, w):
            raise RuntimeError('frame size changed from %s to %s without is_first_frame=True'
                               % (tuple(self.fw_feat.shape[:2]), (h, w)))
        with ops.on_stream(torch.cuda.current_stream()):
            def flow(a, b):
                if pa['zero_flow'] is not None:
                    return pa['zero_flow']
                hit = pa['flows'].get((a, b))
                return hit if hit is not None else self.flow(fr[a], fr[b])
            if is_first_frame:
                for i in range(0, ctr):
                    self.prepare_frame(fr[i])                  # no-op when phase A had the hint
            _, feat_up, conf = self._forward_branch(fr, flow, t, h, w, is_first_frame)
            if is_first_frame:                                                      # :292-295
                self.frame_itr_num = 0
            self.frame_itr_num += 1
        pa['fw_up'], pa['conf_fw'] = feat_up, conf
        return pa

    @torch.no_grad()
    def phase_b2(self, pa, want_vis=False):
        """BW/FW fusion + upsampler (RefVSR.py:288-297) of a frame whose phase_b1 has run: independent of the carried state
        and of the other frames."""
        fr, ctr = pa['fr'], pa['t'] // 2
        with ops.on_stream(torch.cuda.current_stream()):
            out = self.compute_up(pa['bw_up'], pa['fw_up'], pa['conf_bw'], pa['conf_fw'], fr[ctr].lr)
            vis = None
            if want_vis:
                vis = collections.OrderedDict()
                vis['conf_map'] = fr[ctr].conf
                vis['conf_map_prop'] = ops.max2(pa['conf_bw'], pa['conf_fw'])
                vis['conf_map_prop_backward'] = pa['conf_bw']
                vis['conf_map_prop_forward'] = pa['conf_fw']
        return out, vis

    @torch.no_grad()
    def phase_b(self, pa, is_first_frame, want_vis=False, after_state=None):
        """The state-dependent rest of forward() = phase_b1 + phase_b2; (phase_a, phase_b) of a frame == forward() of that
        frame.  after_state: callable invoked once the carried state of this frame is final (before the upsampler is
        enqueued)."""
        self.phase_b1(pa, is_first_frame)
        if after_state is not None:
            after_state()
        return self.phase_b2(pa, want_vis)

    def _sample_vis(self, f, conf_bw, conf_fw):
        """The save_sample block of the `vis` samples (RefVSR.py:301-316; identical in RefVSR_IR.py:367-384) for centre frame f."""
        vis = collections.OrderedDict()
        h, w = f.lr.shape[1:]
        hm, wm = (h // (self.cfg.scale // 2), w // (self.cfg.scale // 2)) if self.hd else (h, w)
        gh, gw = (hm // 2, wm // 2) if self.vgg7 else (hm, wm)
        s1, s2 = self.ks // 2, self.ks
        lr_down = ops.bicubic_scale(f.lr, 0.5, clamp01=True)
        ref_down = ops.bicubic_scale(f.ref, 0.5, clamp01=True)
        vis['FW_aa1_fm_ref_aligned'] = ops.block_gather_rgb(ref_down, f.idx, gh, gw, s1, planar=True)
        if s1 > 1:
            fm8 = ops.block_gather_rgb(ref_down, f.idx, gh, gw, s1)
            rgb8 = ops.block_gather_rgb(f.ref, f.idx, gh, gw, s1)
            vis['FW_aa1_ref_aligned'] = ops.unpack_nhwc16(self.aligned_conv(fm8, lr_down, rgb8, 'aa1.align', s1), 3)
        vis['FW_aa2_fm_ref_aligned'] = ops.block_gather_rgb(f.ref, f.idx, gh, gw, s2, planar=True)
        rgb8 = ops.block_gather_rgb(f.ref, f.idx, gh, gw, s2)
        vis['FW_aa2_ref_aligned'] = ops.unpack_nhwc16(self.aligned_conv(rgb8, f.lr, rgb8, 'aa2.align', s2), 3)

        def norm(x):
            x = x - x.min()
            return x / x.max()
        vis['conf_map_norm'] = norm(f.conf)
        vis['conf_map_prop_backward_norm'] = norm(conf_bw)
        vis['conf_map_prop_forward_norm'] = norm(conf_fw)
        vis['conf_map_prop_norm'] = norm(ops.max2(conf_bw, conf_fw))
        return vis

    def _debug_vis(self, fr, t, is_first_frame, range_start, fw_flow_in, flow, conf_bw, conf_fw, save_sample):
        """The `vis` debugging samples of Network.forward (RefVSR.py:219-221,262-263,301-316), is_log only; planar fp32
        [C,H,W] maps.  (The min/max normalisation of the four confidence maps, models/utils.py:23-32, uses torch
        reductions: debug output, not the hot path.)"""
        ctr = t // 2
        vis = collections.OrderedDict()
        if ctr < t - 1:
            vis['BW_LR_next_warp'] = ops.warp_planar(fr[ctr + 1].lr, flow(ctr, ctr + 1))
        fl = flow(ctr, ctr - 1) if ctr > range_start else (None if is_first_frame else fw_flow_in)
        if fl is not None:
            vis['FW_LR_prev_warp'] = ops.warp_planar(fr[ctr - 1].lr, fl)
        if save_sample:
            vis.update(self._sample_vis(fr[ctr], conf_bw, conf_fw))
        return vis

    def _forward_seq(self, lrs, refs, is_first_frame, want_vis=False, frame_ids=None, want_log=False):
        t, h, w = self._check_window(lrs, refs)
        ctr = t // 2
        dev = lrs.device
        if self.max_frame_itr_num is not None and self.frame_itr_num == self.max_frame_itr_num:
            is_first_frame = True                                                   # :168-170
        if not is_first_frame and self.fw_feat is None:
            raise RuntimeError('is_first_frame=False but no forward state is held (first call of a stream '
                               'must pass is_first_frame=True, cf. RefVSR.py:257-258)')
        if not is_first_frame and tuple(self.fw_feat.shape[:2]) != (h, w):
            raise RuntimeError('frame size changed from %s to %s without is_first_frame=True'
                               % (tuple(self.fw_feat.shape[:2]), (h, w)))
        gradio = bool(self.cfg.EVAL.is_gradio)
        zero_flow = torch.zeros((2, h, w), dtype=torch.float32, device=dev) if gradio else None
        fr = self._frames(lrs, refs, frame_ids)
        range_start = 0 if is_first_frame else ctr                                  # :173-176
        # Two-stream overlap (steady state): the forward-branch step depends only on cached per-frame data and
        # the carried state, not on this window's new frame, so it runs on a side stream while the main stream
        # prepares the new frame (matching, encoders, alignment) and walks the backward branch.
        main = torch.cuda.current_stream()
        overlap = (self.overlap and not is_first_frame and fr[ctr].conf is not None)
        # with two streams every flow carries an event and is recorded on both: at a clip end the window repeats its
        # last frame, the two branches then ask for the SAME pair (fr[ctr], fr[ctr+1]) and the second one takes the
        # first one's tensor out of the cache -- from the other stream
        share = (main, self._side_stream(dev)) if overlap else None
        flow = (lambda a, b: zero_flow) if gradio else (lambda a, b: self.flow(fr[a], fr[b], share))   # :183-191
        fw_flow_in = self.fw_flow
        if not gradio and t > 1:
            # every flow this call consumes, in one batched SPyNet pass before the streams fork (cached pairs cost nothing)
            need = [(fr[i], fr[i + 1]) for i in range(ctr, t - 1)] + [(fr[ctr + 1], fr[ctr])]
            if is_first_frame:
                need += [(fr[i], fr[i - 1]) for i in range(1, ctr + 1)]
            self.flows(need, share)
        if overlap:
            for i in range(ctr, t):
                self.pyramid(fr[i])                        # shared by both streams: build on main before the fork
            side = self._side_stream(dev)
            side.wait_stream(main)
            with ops.on_stream(side):
                for x in (self.fw_feat, self.fw_feat_up, self.fw_conf, self.fw_flow):
                    x.record_stream(side)
                fw = self._forward_branch(fr, flow, t, h, w, False)
        for i in range(range_start, t):                                             # :196-204 (+ per-frame RAP parts)
            self.prepare_frame(fr[i])

        bw_up, conf_bw = self._backward_branch(fr, flow, t, h, w)

        # ---- forward branch (:240-283)
        if overlap:
            main.wait_stream(side)
            for x in fw:
                x.record_stream(main)
        else:
            fw = self._forward_branch(fr, flow, t, h, w, is_first_frame)
        feat, feat_up, conf = fw
        out = self.compute_up(bw_up, feat_up, conf_bw, conf, fr[ctr].lr)            # :288-289,297
        if is_first_frame:                                                          # :292-295
            self.frame_itr_num = 0
        self.frame_itr_num += 1
        vis = None
        if want_vis:                                                                # :318-322
            vis = collections.OrderedDict()
            vis['conf_map'] = fr[ctr].conf
            vis['conf_map_prop'] = ops.max2(conf_bw, conf)
            vis['conf_map_prop_backward'] = conf_bw
            vis['conf_map_prop_forward'] = conf
        if want_log:
            dbg = self._debug_vis(fr, t, is_first_frame, range_start, fw_flow_in, flow, conf_bw, conf, want_vis)
            return out, (vis, dbg)
        return out, vis
