"""PyTorch-ROCm custom ops of the RefVSR hot path: `torch.ops.refvsr.*`.

SURVEY.md 8(b) / BASELINE.json `north_star` ask for the kernels to be callable "from Python via PyTorch-ROCm custom ops".
The C-ABI (include/refvsr_hip.h, bound by refvsr_amd/hip.py) stays the boundary; this module registers the tensor-level
entry points of refvsr_amd/ops.py with `torch.library` on top of it, so that they

  * show up in the dispatcher as `refvsr::<op>` (device type cuda = HIP on ROCm), run on the current HIP stream and
    raise `RuntimeError` with the library's message on bad arguments,
  * carry shape / dtype inference (`register_fake`), i.e. they trace under `torch.compile` / `make_fx` / FakeTensor and
    can be captured in a HIP graph like any other op,
  * are functional (fresh outputs, no aliasing, no mutation).

The engine itself calls the same kernels through refvsr_amd/ops.py directly: one forward issues ~330 launches, and the
dispatcher's per-call cost (5-10 us) would put the host back on the critical path; results are identical
(tests/test_gpu_ops.py::test_torch_library_ops_match_direct_calls).

Op set (argument conventions of ops.py: `planar` = float32 [C,H,W], `nhwc16` = float16 [H,W,Cs]):
  conv_mfma(wpack, bias, meta, src0, src1?, mul?, res?, res_planar?, stride, act, post, out_mode, add_const, clamp_lo, clamp_hi)
  resblock(w1, b1, w2, b2, ksteps, x, act, post)        fused ResidualBlockNoBN / ResBlock
  resblock24_chain_batch(blobs, xs[], act) / conv24_batch(blob, src0[], src1[], mul[], res[], act, post; [] = absent) / warp_batch(xs[], flows[])
                                                         multi-map launches (ABI 11): B maps per launch, one [B, ...] output
  match_argmax(lr_feat, ref_feat) -> (conf, idx)         FeatureMatching GEMM + exact arg-max (attention.py:72-91)
  warp(x_nhwc16, flow) / warp_planar(x, flow)            models/utils.py:35-43
  spynet_level_input(ref, supp, flow_prev?) -> (x8, flow_up)
  block_gather(value_nhwc16, idx, gh, gw, s) / block_gather_rgb(value, idx, gh, gw, s)
  aligned_sample(x_nhwc16, affine, ks)
  resize(x, oh, ow, mode, sy, sx, clamp01) ; pack_nhwc16(x, cs) ; unpack_nhwc16(x, c)
"""
from typing import List, Optional, Tuple

import torch

from . import hip, ops

_LIB = 'refvsr'
_REGISTERED = False


def _conv_out_shape(meta, src0, stride):
    cout, ks, shuffle, f32 = int(meta[0]), int(meta[1]), int(meta[2]), int(meta[3])
    h, w = src0.shape[0], src0.shape[1]
    pad = ks // 2
    return cout, (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1, shuffle, f32


class _PackedConv(object):
    """ops.ConvWeights rebuilt around tensors handed through the dispatcher (no re-upload, no copy)."""

    def __init__(self, wpack, bias, meta):
        self.wpack, self.bias = wpack, bias
        self.cout, self.ksize, shuffle, f32, self.ksteps, self.mt = [int(v) for v in meta[:6]]
        self.shuffle, self.f32 = bool(shuffle), bool(f32)
        self.cpads = [int(v) for v in meta[6:] if int(v) > 0]
        d = self.desc = hip.RefvsrConv()
        d.wpack, d.bias = wpack.data_ptr(), bias.data_ptr()
        d.cout, d.mt_per_block, d.ksteps, d.ksize, d.f32 = self.cout, self.mt, self.ksteps, self.ksize, int(self.f32)
        self.odtype = torch.float32 if self.f32 else torch.float16
        self.raw = self.blob24 = None          # the generic kernel: the specialised ones are their own ops (conv24, conv_shuffle2, resblock24_chain)


def conv_meta(cw):
    """Integer descriptor of packed conv weights for torch.ops.refvsr.conv_mfma: [cout, ksize, shuffle, f32, ksteps, mt, cpad0, cpad1]."""
    cp = list(cw.cpads) + [0, 0]
    return [cw.cout, cw.ksize, int(cw.shuffle), int(cw.f32), cw.ksteps, cw.mt, cp[0], cp[1]]


def register():
    """Idempotent; called on `import refvsr_amd.torch_ops`."""
    global _REGISTERED
    if _REGISTERED:
        return
    _REGISTERED = True
    op = lambda name, **kw: torch.library.custom_op('%s::%s' % (_LIB, name), mutates_args=(), device_types='cuda', **kw)

    @op('conv_mfma')
    def conv_mfma(wpack: torch.Tensor, bias: torch.Tensor, meta: List[int], src0: torch.Tensor, src1: Optional[torch.Tensor],
                  mul: Optional[torch.Tensor], res: Optional[torch.Tensor], res_planar: Optional[torch.Tensor], stride: int,
                  act: float, post: float, planar_out: bool, add_const: float, clamp_lo: float, clamp_hi: float) -> torch.Tensor:
        cw = _PackedConv(wpack, bias, list(meta))
        clamp = (clamp_lo, clamp_hi) if clamp_lo < clamp_hi else None
        return ops.conv(cw, src0, src1, stride=stride, act=act, mul=mul, res=res, post=post, planar_out=planar_out,
                        res_planar=res_planar, add_const=add_const, clamp=clamp)

    @conv_mfma.register_fake
    def _(wpack, bias, meta, src0, src1, mul, res, res_planar, stride, act, post, planar_out, add_const, clamp_lo, clamp_hi):
        cout, ho, wo, shuffle, f32 = _conv_out_shape(list(meta), src0, stride)
        if planar_out:
            return src0.new_empty((cout, ho, wo), dtype=torch.float32)
        ru = lambda a, b: (a + b - 1) // b * b          # channel stride of the map, as ops.conv allocates it (C = 36 -> 40)
        if shuffle:
            return src0.new_empty((2 * ho, 2 * wo, ru(cout // 4, 8)), dtype=torch.float16)
        return src0.new_empty((ho, wo, ru(cout, 4 if f32 else 8)), dtype=torch.float32 if f32 else torch.float16)

    @op('resblock')
    def resblock(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, ksteps: int, x: torch.Tensor,
                 act: float, post: float) -> torch.Tensor:
        c = x.shape[2]
        mk = lambda w, b: _PackedConv(w, b, [c, 3, 0, 0, ksteps, (c + 15) // 16, c, 0])
        return ops.resblock(mk(w1, b1), mk(w2, b2), x, act=act, post=post)

    @resblock.register_fake
    def _(w1, b1, w2, b2, ksteps, x, act, post):
        return torch.empty_like(x)

    @op('conv24')
    def conv24(blob: torch.Tensor, src0: torch.Tensor, src1: Optional[torch.Tensor], mul: Optional[torch.Tensor],
               res: Optional[torch.Tensor], act: float, post: float) -> torch.Tensor:
        h, w, c0 = src0.shape
        out = torch.empty((h, w, 24), dtype=torch.float16, device=src0.device)
        hip.check(hip.lib().refvsr_conv24(ops._ptr(src0), c0, ops._ptr(src1), 0 if src1 is None else src1.shape[2], h, w, ops._ptr(blob), act,
                                          ops._ptr(mul), ops._ptr(res), post, ops._ptr(out), ops._stream()), 'conv24')
        return out

    @conv24.register_fake
    def _(blob, src0, src1, mul, res, act, post):
        return src0.new_empty((src0.shape[0], src0.shape[1], 24), dtype=torch.float16)

    @op('conv_shuffle2')
    def conv_shuffle2(blobs: torch.Tensor, src: torch.Tensor, act: float) -> torch.Tensor:
        h, w, c = src.shape
        out = torch.empty((2 * h, 2 * w, c), dtype=torch.float16, device=src.device)
        hip.check(hip.lib().refvsr_conv_shuffle2(ops._ptr(src), c, h, w, ops._ptr(blobs), act, ops._ptr(out), ops._stream()), 'conv_shuffle2')
        return out

    @conv_shuffle2.register_fake
    def _(blobs, src, act):
        return src.new_empty((2 * src.shape[0], 2 * src.shape[1], src.shape[2]), dtype=torch.float16)

    @op('resblock24_chain')
    def resblock24_chain(blobs: torch.Tensor, x: torch.Tensor, act: float) -> torch.Tensor:
        class _Ch(object):
            pass
        ch = _Ch()
        ch.n, ch.blobs, ch.stride = blobs.shape[0], blobs, blobs.shape[1]
        return ops.resblock24_chain(ch, x, act)

    @resblock24_chain.register_fake
    def _(blobs, x, act):
        return torch.empty_like(x)

    # ---- multi-map launches (ABI 11): B maps of one geometry behind one launch per layer.  Tensor[] in, one [B, ...] tensor out.
    @op('resblock24_chain_batch')
    def resblock24_chain_batch(blobs: torch.Tensor, xs: List[torch.Tensor], act: float) -> torch.Tensor:
        class _Ch(object):
            pass
        ch = _Ch()
        ch.n, ch.blobs, ch.stride = blobs.shape[0], blobs, blobs.shape[1]
        return ops.resblock24_chain_b(ch, list(xs), act)

    @resblock24_chain_batch.register_fake
    def _(blobs, xs, act):
        return xs[0].new_empty((len(xs),) + tuple(xs[0].shape))

    @op('conv24_batch')
    def conv24_batch(blob: torch.Tensor, src0: List[torch.Tensor], src1: List[torch.Tensor], mul: List[torch.Tensor],
                     res: List[torch.Tensor], act: float, post: float) -> torch.Tensor:
        # (an EMPTY list = operand absent: the dispatcher's schema language has no optional tensor list)
        B = len(src0)
        h, w, c0 = src0[0].shape
        c1 = src1[0].shape[2] if len(src1) else 0
        out = torch.empty((B, h, w, 24), dtype=torch.float16, device=src0[0].device)
        pa = lambda l_: ops._parr(list(l_)) if len(l_) else None
        hip.check(hip.lib().refvsr_conv24_batch(pa(src0), c0, pa(src1), c1, B, h, w, ops._ptr(blob), act, pa(mul), pa(res), post,
                                                ops._parr(list(out)), ops._stream()), 'conv24_batch')
        return out

    @conv24_batch.register_fake
    def _(blob, src0, src1, mul, res, act, post):
        return src0[0].new_empty((len(src0), src0[0].shape[0], src0[0].shape[1], 24), dtype=torch.float16)

    @op('warp_batch')
    def warp_batch(xs: List[torch.Tensor], flows: List[torch.Tensor]) -> torch.Tensor:
        return ops.warp_nhwc16_b(list(xs), list(flows))

    @warp_batch.register_fake
    def _(xs, flows):
        return xs[0].new_empty((len(xs), flows[0].shape[1], flows[0].shape[2], xs[0].shape[2]))

    @op('match_argmax')
    def match_argmax(lr_feat: torch.Tensor, ref_feat: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        lr_rows, inv_lr, lr_lo = ops.match_patches(lr_feat, hip.MATCH_COLBLOCK, want_lo=True)
        ref_rows, inv_ref, ref_lo = ops.match_patches(ref_feat, hip.MATCH_ROWCHUNK, want_lo=True)
        n_lr, n_ref = lr_feat.shape[1] * lr_feat.shape[2], ref_feat.shape[1] * ref_feat.shape[2]
        cand, cval = ops.match_top2(ref_rows, n_ref, lr_rows, n_lr, 1)
        conf, idx, _ = ops.match_refine(lr_feat, ref_feat, inv_lr, inv_ref, cand, cval, ops.MATCH_EXACT_MARGIN, (lr_rows, lr_lo), (ref_rows, ref_lo))
        return conf.view(1, lr_feat.shape[1], lr_feat.shape[2]), idx

    @match_argmax.register_fake
    def _(lr_feat, ref_feat):
        h, w = lr_feat.shape[1], lr_feat.shape[2]
        return lr_feat.new_empty((1, h, w)), lr_feat.new_empty((h * w,), dtype=torch.int32)

    @op('warp')
    def warp(x: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
        return ops.warp_nhwc16(x, flow)

    @warp.register_fake
    def _(x, flow):
        return x.new_empty((flow.shape[1], flow.shape[2], x.shape[2]))

    @op('warp_planar')
    def warp_planar(x: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
        return ops.warp_planar(x, flow)

    @warp_planar.register_fake
    def _(x, flow):
        return x.new_empty((x.shape[0], flow.shape[1], flow.shape[2]))

    @op('spynet_level_input')
    def spynet_level_input(ref: torch.Tensor, supp: torch.Tensor, flow_prev: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        return ops.spynet_level_input(ref, supp, flow_prev)

    @spynet_level_input.register_fake
    def _(ref, supp, flow_prev):
        h, w = ref.shape[1], ref.shape[2]
        return ref.new_empty((h, w, 8), dtype=torch.float16), ref.new_empty((2, h, w))

    @op('block_gather')
    def block_gather(value: torch.Tensor, idx: torch.Tensor, gh: int, gw: int, s: int) -> torch.Tensor:
        return ops.block_gather_nhwc16(value, idx, gh, gw, s)

    @block_gather.register_fake
    def _(value, idx, gh, gw, s):
        return value.new_empty((gh * s, gw * s, value.shape[2]))

    @op('block_gather_rgb')
    def block_gather_rgb(value: torch.Tensor, idx: torch.Tensor, gh: int, gw: int, s: int) -> torch.Tensor:
        return ops.block_gather_rgb(value, idx, gh, gw, s)

    @block_gather_rgb.register_fake
    def _(value, idx, gh, gw, s):
        return value.new_empty((gh * s, gw * s, 8), dtype=torch.float16)

    @op('aligned_sample')
    def aligned_sample(x: torch.Tensor, affine: torch.Tensor, ks: int) -> torch.Tensor:
        return ops.aligned_sample(x, affine, ks)

    @aligned_sample.register_fake
    def _(x, affine, ks):
        return torch.empty_like(x)

    @op('resize')
    def resize(x: torch.Tensor, oh: int, ow: int, mode: int, sy: float, sx: float, clamp01: bool) -> torch.Tensor:
        return ops.resize(x, (oh, ow), mode, (sy, sx), clamp01=clamp01)

    @resize.register_fake
    def _(x, oh, ow, mode, sy, sx, clamp01):
        return x.new_empty((x.shape[0], oh, ow))

    @op('pack_nhwc16')
    def pack_nhwc16(x: torch.Tensor, cs: int) -> torch.Tensor:
        return ops.pack_nhwc16(x, cs)

    @pack_nhwc16.register_fake
    def _(x, cs):
        return x.new_empty((x.shape[1], x.shape[2], cs), dtype=torch.float16)

    @op('unpack_nhwc16')
    def unpack_nhwc16(x: torch.Tensor, c: int) -> torch.Tensor:
        return ops.unpack_nhwc16(x, c)

    @unpack_nhwc16.register_fake
    def _(x, c):
        return x.new_empty((c, x.shape[0], x.shape[1]), dtype=torch.float32)


OP_NAMES = ('conv_mfma', 'conv24', 'resblock', 'resblock24_chain', 'resblock24_chain_batch', 'conv24_batch', 'warp_batch', 'match_argmax', 'warp', 'warp_planar', 'spynet_level_input', 'block_gather',
            'block_gather_rgb', 'aligned_sample', 'resize', 'pack_nhwc16', 'unpack_nhwc16')

register()
