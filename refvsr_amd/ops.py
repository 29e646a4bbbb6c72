"""Tensor-level wrappers over the C-ABI (refvsr_amd/hip.py).  torch is used only for device memory
(caching allocator) and the current HIP stream; every computation is a kernel of librefvsr_hip.so.

Layouts: `planar` = float32 [C,H,W];  `nhwc16` = float16 [H,W,Cs] (Cs % 8 == 0).
"""
import ctypes as C
import math
import os

import torch

from . import hip
from .knobs import env_flag
from .hip import (OUT_NHWC16, OUT_NHWC16_SHUFFLE2, OUT_PLANAR32, RS_BICUBIC, RS_BILINEAR,  # noqa: F401
                  RS_BILINEAR_AC, RS_NEAREST)


_STREAM = None        # cached hipStream_t of the stream selected with on_stream(); None -> ask torch every time


def _stream():
    if _STREAM is not None:
        return _STREAM
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class on_stream(object):
    """`with ops.on_stream(s):` == `with torch.cuda.stream(s):` + caches the raw stream handle for the kernel
    launches inside (torch.cuda.current_stream() costs ~8 us of host time per call, x ~350 launches per frame)."""

    def __init__(self, stream):
        self.stream = stream
        self.ctx = torch.cuda.stream(stream)

    def __enter__(self):
        global _STREAM
        self.prev = _STREAM
        self.ctx.__enter__()
        _STREAM = C.c_void_p(self.stream.cuda_stream)
        return self.stream

    def __exit__(self, *exc):
        global _STREAM
        _STREAM = self.prev
        return self.ctx.__exit__(*exc)


def num_cus():
    """CU count of the current device as the library sees it."""
    return int(hip.lib().refvsr_num_cus())


class CuStream(torch.cuda.ExternalStream):
    """A HIP stream restricted to the CUs [first_cu, first_cu + n_cus) (refvsr_stream_create_cu_range: both multiples of 8 = an equal
    share of every XCD).  The persistent launchers of the library size their grids for n_cus CUs on it.  A torch stream in every other
    respect (events, wait_stream, record_stream, `with ops.on_stream(s)`).  The HIP stream lives as long as the process: engines keep
    their streams for their whole life and the allocator may hold record_stream references to it."""

    def __new__(cls, first_cu, n_cus, device=None):
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        out = C.c_void_p(0)
        with torch.cuda.device(dev):
            hip.check(hip.lib().refvsr_stream_create_cu_range(int(first_cu), int(n_cus), C.byref(out)), 'stream_create_cu_range')
        self = super().__new__(cls, out.value, device=dev)
        self.first_cu, self.n_cus = int(first_cu), int(n_cus)
        return self


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _planar(t, c=None):
    assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.is_contiguous(), \
        'expected contiguous cuda float32 [C,H,W], got %s %s' % (t.dtype, tuple(t.shape))
    if c is not None:
        assert t.shape[0] == c
    return t


def _nhwc(t, f32=False):
    if f32:
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.is_contiguous() and t.shape[2] % 4 == 0, \
            'expected contiguous cuda float32 [H,W,C%%4==0], got %s %s' % (t.dtype, tuple(t.shape))
        return t
    assert t.is_cuda and t.dtype == torch.float16 and t.dim() == 3 and t.is_contiguous() and t.shape[2] % 8 == 0, \
        'expected contiguous cuda float16 [H,W,C%%8==0], got %s %s' % (t.dtype, tuple(t.shape))
    return t


def _round_up(a, b):
    return (a + b - 1) // b * b


def _farr(vals):
    if vals is None:
        return None
    return (C.c_float * len(vals))(*[float(v) for v in vals])


CONV24 = not env_flag('REFVSR_NO_CONV24')      # A/B knob: the generic conv kernel for the conv24 shapes as well


class ConvWeights(object):
    """Packed weights of one conv on the device (see packing.pack_conv)."""
    __slots__ = ('wpack', 'bias', 'cout', 'ksteps', 'mt', 'ksize', 'cpads', 'shuffle', 'f32', 'hi_only', 'desc', 'odtype', 'raw', 'blob24')

    def __init__(self, pk, device):
        self.wpack = pk['wpack'].to(device).contiguous()
        self.bias = pk['bias'].to(device).contiguous()
        self.cout, self.ksteps, self.mt, self.ksize = pk['cout'], pk['ksteps'], pk['mt'], pk['ksize']
        self.cpads, self.shuffle, self.f32 = pk['cpads'], pk['shuffle'], bool(pk.get('f32', False))
        self.hi_only = bool(pk.get('hi_only', False))      # plain fp16 weights (descriptor weight mode 2)
        # launch descriptor with the per-weight fields filled once (the C side copies it at every call)
        d = self.desc = hip.RefvsrConv()
        d.wpack, d.bias = self.wpack.data_ptr(), self.bias.data_ptr()
        d.cout, d.mt_per_block, d.ksteps, d.ksize = self.cout, self.mt, self.ksteps, self.ksize
        d.f32 = 2 if self.hi_only else int(self.f32)
        self.odtype = torch.float32 if self.f32 else torch.float16
        self.raw = pk.get('raw')      # (weight, bias) fp32 cpu tensors when the packer kept them (repacking for specialised kernels)
        # 3x3 convs with 24 / 48 output channels of the supported input shapes also carry the blob of the specialised kernel
        # (conv24.hip; the attribute keeps its first name)
        self.blob24 = None
        if self.raw is not None and pk.get('src_channels') is not None and CONV24 and not self.hi_only:
            from .packing import conv24_ok, conv_shuffle2_ok, pack_conv24, pack_conv_shuffle2
            two48 = self.cout == 48 and len(pk['src_channels']) == 2     # 48 + 48 -> 48 as two channel halves (A/B knob)
            if (conv24_ok(tuple(self.raw[0].shape), pk['src_channels'], self.shuffle, self.f32) and not (self.cout == 32 and env_flag('REFVSR_NO_CONV32'))
                    and not (two48 and env_flag('REFVSR_NO_CONV48X2'))):
                self.blob24 = pack_conv24(self.raw[0], self.raw[1], pk['src_channels']).to(device).contiguous()
            elif self.shuffle and conv_shuffle2_ok(tuple(self.raw[0].shape), pk['src_channels'], self.f32) and not env_flag('REFVSR_NO_CONV_SHUFFLE2'):
                self.blob24 = pack_conv_shuffle2(self.raw[0], self.raw[1]).to(device).contiguous()     # refvsr_conv_shuffle2


def conv(cw, src0, src1=None, stride=1, pad=None, act=1.0, mul=None, res=None, post=1.0,
         planar_out=False, res_planar=None, add_const=0.0, clamp=None, batch=None):
    """refvsr_conv_mfma.  Returns nhwc16 [ho,wo,cout] (or [2ho,2wo,cout/4] for pixel-shuffle weights),
    or planar fp32 [cout,ho,wo] when planar_out.  For f32-packed weights all maps are fp32 HWC.
    batch = B: src0 is a contiguous batch [B,h,w,c] (res_planar [B,cout,h,w]) of images sharing the weights: ONE launch
    (RefvsrConv.batch), output [B,...]; image b == conv of image b alone, bit for bit."""
    if batch is not None:
        return _conv_batch(cw, src0, int(batch), stride, pad, act, post, planar_out, res_planar, add_const, clamp,
                           src1, mul, res)
    f32 = cw.f32
    _nhwc(src0, f32)
    if src1 is not None:
        _nhwc(src1, f32)
    c0 = src0.shape[2]
    c1 = src1.shape[2] if src1 is not None else 0
    h, w = src0.shape[:2]
    assert src1 is None or src1.shape[:2] == src0.shape[:2]
    assert [c0] + ([c1] if src1 is not None else []) == list(cw.cpads), \
        'conv input channels %s do not match packed weights %s' % ([c0, c1], cw.cpads)
    k = cw.ksize
    if pad is None:
        pad = k // 2
    co_ = cw.cout
    if (cw.shuffle and cw.blob24 is not None and stride == 1 and pad == 1 and not planar_out and res_planar is None and
            src1 is None and mul is None and res is None and 0.0 <= act <= 1.0 and post == 1.0 and h * w * co_ * 2 < 2 ** 31):
        # C -> 4 C conv + pixel shuffle on the compile-time-specialised kernel (csrc/conv24.hip, SHUF variant)
        out = torch.empty((2 * h, 2 * w, c0), dtype=torch.float16, device=src0.device)
        hip.check(hip.lib().refvsr_conv_shuffle2(_ptr(src0), c0, h, w, _ptr(cw.blob24), act, _ptr(out), _stream()), 'conv_shuffle2')
        return out
    if (cw.blob24 is not None and not cw.shuffle and stride == 1 and pad == 1 and not planar_out and res_planar is None and
            0.0 <= act <= 1.0 and 0.0 <= post <= 1.0 and (mul is None or mul.shape[2] == co_) and (res is None or res.shape[2] == co_) and
            h * w * max(co_, c0, c1) * 2 < 2 ** 31):         # (32-bit element offsets in the specialised kernels: 8K HR maps go generic)
        # compile-time-specialised kernel (24 | 32 | 48 output channels, 3x3): csrc/conv24.hip
        for m_ in (mul, res):
            if m_ is not None:
                _nhwc(m_)
                assert tuple(m_.shape[:2]) == (h, w)
        out = torch.empty((h, w, co_), dtype=torch.float16, device=src0.device)
        fn = {24: hip.lib().refvsr_conv24, 32: hip.lib().refvsr_conv32, 48: hip.lib().refvsr_conv48}[co_]
        hip.check(fn(_ptr(src0), c0, _ptr(src1), c1, h, w, _ptr(cw.blob24), act, _ptr(mul), _ptr(res), post, _ptr(out), _stream()), 'conv%d' % co_)
        return out
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    d = cw.desc
    d.src0, d.c0, d.src1, d.c1 = src0.data_ptr(), c0, (src1.data_ptr() if src1 is not None else None), c1
    d.h_in, d.w_in, d.h_out, d.w_out = h, w, ho, wo
    d.stride, d.pad = stride, pad
    d.act_slope, d.post_slope = act, post
    if mul is not None:
        _nhwc(mul, f32)
        assert tuple(mul.shape[:2]) == (ho, wo)
        d.mul, d.mul_c = mul.data_ptr(), mul.shape[2]
    else:
        d.mul, d.mul_c = None, 0
    if res is not None:
        _nhwc(res, f32)
        assert tuple(res.shape[:2]) == (ho, wo)
        d.res, d.res_c = res.data_ptr(), res.shape[2]
    else:
        d.res, d.res_c = None, 0
    d.res_planar, d.add_const, d.clamp_lo, d.clamp_hi = None, 0.0, 0.0, 0.0
    if planar_out:
        out = torch.empty((cw.cout, ho, wo), dtype=torch.float32, device=src0.device)
        d.out_mode, d.out_c = OUT_PLANAR32, 0
        if res_planar is not None:
            _planar(res_planar, cw.cout)
            assert tuple(res_planar.shape[1:]) == (ho, wo)
            d.res_planar = res_planar.data_ptr()
        d.add_const = add_const
        if clamp is not None:
            d.clamp_lo, d.clamp_hi = clamp
    elif cw.shuffle:
        co = _round_up(cw.cout // 4, 8)             # channel stride of the map (padding channels are written as zeros)
        out = torch.empty((2 * ho, 2 * wo, co), dtype=torch.float16, device=src0.device)
        d.out_mode, d.out_c = OUT_NHWC16_SHUFFLE2, co
    else:
        co = _round_up(cw.cout, 4 if f32 else 8)
        out = torch.empty((ho, wo, co), dtype=cw.odtype, device=src0.device)
        d.out_mode, d.out_c = OUT_NHWC16, co
    d.out = out.data_ptr()
    rc = hip.lib().refvsr_conv_mfma(C.byref(d), _stream())
    hip.check(rc, 'conv_mfma')
    return out


def _conv_batch(cw, src0, B, stride, pad, act, post, planar_out, res_planar, add_const, clamp, src1, mul, res):
    assert src1 is None and mul is None and res is None and not cw.shuffle, 'batched conv: single source, no mul / res'
    assert src0.dim() == 4 and src0.shape[0] == B and src0.is_contiguous() and B >= 1
    f32 = cw.f32
    _nhwc(src0[0], f32)
    h, w, c0 = src0.shape[1:]
    assert [c0] == list(cw.cpads), 'conv input channels %s do not match packed weights %s' % ([c0], cw.cpads)
    k = cw.ksize
    if pad is None:
        pad = k // 2
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    d = cw.desc
    d.src0, d.c0, d.src1, d.c1 = src0.data_ptr(), c0, None, 0
    d.h_in, d.w_in, d.h_out, d.w_out = h, w, ho, wo
    d.stride, d.pad = stride, pad
    d.act_slope, d.post_slope = act, post
    d.mul, d.mul_c, d.res, d.res_c = None, 0, None, 0
    d.res_planar, d.add_const, d.clamp_lo, d.clamp_hi = None, 0.0, 0.0, 0.0
    esz = 4 if f32 else 2
    d.bs_res_planar = 0
    if planar_out:
        out = torch.empty((B, cw.cout, ho, wo), dtype=torch.float32, device=src0.device)
        d.out_mode, d.out_c = OUT_PLANAR32, 0
        if res_planar is not None:
            assert res_planar.is_cuda and res_planar.dtype == torch.float32 and res_planar.is_contiguous() and \
                tuple(res_planar.shape) == (B, cw.cout, ho, wo)
            d.res_planar = res_planar.data_ptr()
            d.bs_res_planar = cw.cout * ho * wo * 4
        d.add_const = add_const
        if clamp is not None:
            d.clamp_lo, d.clamp_hi = clamp
        d.bs_out = cw.cout * ho * wo * 4
    else:
        co = _round_up(cw.cout, 4 if f32 else 8)
        out = torch.empty((B, ho, wo, co), dtype=cw.odtype, device=src0.device)
        d.out_mode, d.out_c = OUT_NHWC16, co
        d.bs_out = ho * wo * co * esz
    d.out = out.data_ptr()
    d.batch, d.bs_src0, d.bs_src1 = B, h * w * c0 * esz, 0
    try:
        hip.check(hip.lib().refvsr_conv_mfma(C.byref(d), _stream()), 'conv_mfma')
    finally:
        d.batch = 0                                  # the descriptor is shared with the single-image calls
    return out


_WAVES_SET = False


def resblock_fits(c):
    """Channel counts of the runtime-generic fused block (resblock_lean.hip: C in {8, 16, 24, 32})."""
    return bool(hip.lib().refvsr_resblock_lean_fits(int(c)))


def _apply_resblock_knobs():
    global _WAVES_SET
    if not _WAVES_SET:                     # A/B knob of the lean kernel's workgroup shape (default 8 waves)
        _WAVES_SET = True
        if os.environ.get('REFVSR_RESBLOCK_WAVES'):
            hip.check(hip.lib().refvsr_set_resblock_waves(int(os.environ['REFVSR_RESBLOCK_WAVES'])), 'set_resblock_waves')


def resblock(cw1, cw2, x, act, post=1.0):
    """refvsr_resblock_lean: post(x + conv2(act(conv1(x)))) in one launch (3x3, C->C)."""
    _nhwc(x)
    h, w, c = x.shape
    assert cw1.cpads == [c] and cw2.cpads == [c] and cw1.cout == c and cw2.cout == c and cw1.ksize == 3
    assert not cw1.f32 and not cw1.shuffle and cw1.wpack.shape[0] == 1
    out = torch.empty_like(x)
    _apply_resblock_knobs()
    hip.check(hip.lib().refvsr_resblock_lean(_ptr(x), c, h, w, _ptr(cw1.wpack), _ptr(cw1.bias), _ptr(cw2.wpack),
                                             _ptr(cw2.bias), cw1.ksteps, act, post, _ptr(out), _stream()), 'resblock_lean')
    return out


class ResblockChain(object):
    """Pointer tables of a run of fused blocks (built once per run of packed weights, reused by every call)."""

    def __init__(self, pairs):
        self.pairs = list(pairs)
        n = self.n = len(self.pairs)
        c1 = self.pairs[0][0]
        self.c, self.ksteps = c1.cout, c1.ksteps
        for a, b in self.pairs:
            assert a.cpads == [self.c] and b.cpads == [self.c] and a.cout == self.c and b.cout == self.c and a.ksize == 3
            assert not a.f32 and not a.shuffle and a.wpack.shape[0] == 1 and a.ksteps == self.ksteps
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        self.w1, self.b1 = arr([a.wpack for a, _ in self.pairs]), arr([a.bias for a, _ in self.pairs])
        self.w2, self.b2 = arr([b.wpack for _, b in self.pairs]), arr([b.bias for _, b in self.pairs])


def resblock_chain(chain, x, act, post=1.0):
    """refvsr_resblock_chain: chain.n fused blocks behind ONE library call (same launches and results as chain.n calls of
    resblock(); two scratch maps instead of n - 1 intermediates).  """
    _nhwc(x)
    h, w, c = x.shape
    assert c == chain.c
    out = torch.empty_like(x)
    s0 = torch.empty_like(x) if chain.n >= 2 else None
    s1 = torch.empty_like(x) if chain.n >= 3 else None
    hip.check(hip.lib().refvsr_resblock_chain(_ptr(x), c, h, w, chain.n, chain.w1, chain.b1, chain.w2, chain.b2, chain.ksteps,
                                              act, post, _ptr(s0), _ptr(s1), _ptr(out), _stream()), 'resblock_chain')
    return out


class Resblock24Chain(object):
    """A run of 24-channel fused blocks for refvsr_resblock24_chain: one device buffer [n, 43264] of per-block blobs
    (packing.pack_resblock24).  pairs: [(conv1, conv2)] ConvWeights whose packer kept the raw fp32 weights, or
    [((w1, b1), (w2, b2))] raw tensors."""

    def __init__(self, pairs, device):
        from .packing import pack_resblock24
        blobs = []
        for a, b in pairs:
            (w1, b1), (w2, b2) = (a.raw if isinstance(a, ConvWeights) else a), (b.raw if isinstance(b, ConvWeights) else b)
            blobs.append(pack_resblock24(w1, b1, w2, b2))
        self.n = len(blobs)
        self.blobs = torch.stack(blobs, 0).to(device).contiguous()
        self.stride = self.blobs.shape[1]
        assert self.stride == hip.RESBLOCK24_BLOB_BYTES and self.blobs.data_ptr() % 16 == 0


_RB24_WAVES_SET = False


def resblock24_chain(chain, x, act):
    """refvsr_resblock24_chain: chain.n fused 24-channel blocks x <- x + conv2(act(conv1 x)) behind one library call."""
    global _RB24_WAVES_SET
    if not _RB24_WAVES_SET:                # A/B knob of the workgroup shape (default 8 waves)
        _RB24_WAVES_SET = True
        if os.environ.get('REFVSR_RESBLOCK24_WAVES'):
            hip.check(hip.lib().refvsr_set_resblock24_waves(int(os.environ['REFVSR_RESBLOCK24_WAVES'])), 'set_resblock24_waves')
        if os.environ.get('REFVSR_RB24_STORE'):           # A/B knob of the output store path (0 | 1, refvsr_set_resblock24_store)
            hip.check(hip.lib().refvsr_set_resblock24_store(int(os.environ['REFVSR_RB24_STORE'])), 'set_resblock24_store')
    _nhwc(x)
    h, w, c = x.shape
    assert c == 24
    out = torch.empty_like(x)
    s0 = torch.empty_like(x) if chain.n >= 2 else None
    s1 = torch.empty_like(x) if chain.n >= 3 else None
    hip.check(hip.lib().refvsr_resblock24_chain(_ptr(x), h, w, chain.n, _ptr(chain.blobs), chain.stride, act, _ptr(s0), _ptr(s1),
                                                _ptr(out), _stream()), 'resblock24_chain')
    return out


class Resblock48Chain(object):
    """A run of 48-channel fused blocks for refvsr_resblock48_chain: one device buffer [n, 172544] of per-block blobs
    (packing.pack_resblock48).  pairs: [(conv1, conv2)] ConvWeights whose packer kept the raw fp32 weights, or raw tensors."""

    def __init__(self, pairs, device):
        from .packing import pack_resblock48
        blobs = []
        for a, b in pairs:
            (w1, b1), (w2, b2) = (a.raw if isinstance(a, ConvWeights) else a), (b.raw if isinstance(b, ConvWeights) else b)
            blobs.append(pack_resblock48(w1, b1, w2, b2))
        self.n = len(blobs)
        self.blobs = torch.stack(blobs, 0).to(device).contiguous()
        self.stride = self.blobs.shape[1]
        assert self.stride == hip.RESBLOCK48_BLOB_BYTES and self.blobs.data_ptr() % 16 == 0


def resblock48_chain(chain, x, act):
    """refvsr_resblock48_chain: chain.n fused 48-channel blocks x <- x + conv2(act(conv1 x)), one launch per block."""
    _nhwc(x)
    h, w, c = x.shape
    assert c == 48
    out = torch.empty_like(x)
    s0 = torch.empty_like(x) if chain.n >= 2 else None
    s1 = torch.empty_like(x) if chain.n >= 3 else None
    hip.check(hip.lib().refvsr_resblock48_chain(_ptr(x), h, w, chain.n, _ptr(chain.blobs), chain.stride, act, _ptr(s0), _ptr(s1),
                                                _ptr(out), _stream()), 'resblock48_chain')
    return out


def resblock_chain_ok(c):
    _apply_resblock_knobs()
    return bool(hip.lib().refvsr_resblock_lean_fits(int(c)))


def conv_direct(x, w, b, stride=1, pad=None, act=1.0, nhwc16_out=False):
    """refvsr_conv_direct_f32 on a planar fp32 map; w fp32 [cout,cin,k,k] on the device."""
    _planar(x)
    cout, cin, k, _ = w.shape
    assert x.shape[0] == cin and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
    if pad is None:
        pad = k // 2
    h, wd = x.shape[1:]
    ho = (h + 2 * pad - k) // stride + 1
    wo = (wd + 2 * pad - k) // stride + 1
    if nhwc16_out:
        assert cout % 8 == 0
        out = torch.empty((ho, wo, cout), dtype=torch.float16, device=x.device)
    else:
        out = torch.empty((cout, ho, wo), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().refvsr_conv_direct_f32(_ptr(x), cin, h, wd, _ptr(w), _ptr(b), cout, k, stride, pad, act,
                                               _ptr(out), int(nhwc16_out), cout if nhwc16_out else 0, _stream()),
              'conv_direct_f32')
    return out


def conf_alpha(conf_a, conf_b, up, w0, b0, cw, slope0=0.2, slope1=0.2, want_max=False):
    """refvsr_conf_alpha: conv_{16->C}(lrelu(conv_{2->16}(P))) with P = cat[conf_a, conf_b] (up = 1) or its clamped bicubic x2
    up-sampling (up = 2), one launch.  conf_a / conf_b planar fp32 [1,h,w]; w0 / b0 the 2 -> 16 conv (fp32, device); cw the packed
    16 -> C conv (ConvWeights with a conv24 / conv48 blob).  Returns alpha [up h, up w, C] (and max(conf_a, conf_b) [1,h,w])."""
    _planar(conf_a, 1)
    _planar(conf_b, 1)
    assert conf_a.shape == conf_b.shape and cw.blob24 is not None and cw.cpads == [16] and cw.cout in (24, 48)
    assert tuple(w0.shape) == (16, 2, 3, 3) and w0.is_cuda and w0.dtype == torch.float32 and w0.is_contiguous() and b0.numel() == 16
    h, w = conf_a.shape[1:]
    alpha = torch.empty((up * h, up * w, cw.cout), dtype=torch.float16, device=conf_a.device)
    cmax = torch.empty_like(conf_a) if want_max else None
    hip.check(hip.lib().refvsr_conf_alpha(_ptr(conf_a), _ptr(conf_b), h, w, up, _ptr(w0), _ptr(b0), slope0, _ptr(cw.blob24), cw.cout,
                                          slope1, _ptr(alpha), _ptr(cmax), _stream()), 'conf_alpha')
    return (alpha, cmax) if want_max else alpha


RESULT_DTYPES = {'float32': (torch.float32, hip.RESULT_F32), 'float16': (torch.float16, hip.RESULT_F16), 'uint8': (torch.uint8, hip.RESULT_U8)}


def result_format(name):
    """(torch dtype, REFVSR_RESULT_*) of config.result_dtype ('float32' | 'float16' | 'uint8'; None = 'float32')."""
    key = str(name or 'float32').replace('torch.', '')
    if key not in RESULT_DTYPES:
        raise ValueError("result_dtype must be 'float32', 'float16' or 'uint8', got %r" % (name,))
    return RESULT_DTYPES[key]


def convert_result(x, result_dtype):
    """fp32 result in [0, 1] -> fp16 | uint8 = rint(255 x) (refvsr_convert_result: what the fused heads store directly)."""
    dt, fmt = result_format(result_dtype)
    if fmt == hip.RESULT_F32:
        return x
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=dt, device=x.device)
    hip.check(hip.lib().refvsr_convert_result(_ptr(x), x.numel(), fmt, _ptr(out), _stream()), 'convert_result')
    return out


def conv_last(blob, src, base_lr, result_dtype=None):
    """refvsr_conv_last: clamp(conv3x3_{C->3}(src) + bias + clamp01(bicubic(base_lr)), 0, 1) -> planar [3, h, w] in one launch (fp32, or
    result_dtype = 'float16' | 'uint8': REFVSR_RESULT_*).
    blob: packing.pack_conv_last on the device; src nhwc16 [h, w, C]; base_lr planar fp32 [3, h / s, w / s]."""
    _nhwc(src)
    _planar(base_lr, 3)
    h, w, c = src.shape
    bh, bw = base_lr.shape[1:]
    dt, fmt = result_format(result_dtype)
    out = torch.empty((3, h, w), dtype=dt, device=src.device)
    hip.check(hip.lib().refvsr_conv_last_fmt(_ptr(src), c, h, w, _ptr(blob), _ptr(base_lr), bh, bw, _ptr(out), fmt, _stream()), 'conv_last')
    return out


def conv_hr_last(blob, src, base_lr, act=0.1, result_dtype=None):
    """refvsr_conv_hr_last: clamp(conv_last(lrelu(conv_hr(src))) + clamp01(bicubic(base_lr)), 0, 1) -> planar [3, h, w], one launch
    (mid_channels = 24; fp32, or result_dtype = 'float16' | 'uint8').  blob: packing.pack_conv_hr_last on the device."""
    _nhwc(src)
    _planar(base_lr, 3)
    h, w, c = src.shape
    assert c == 24 and blob.numel() == hip.RESBLOCK24_BLOB_BYTES
    bh, bw = base_lr.shape[1:]
    dt, fmt = result_format(result_dtype)
    out = torch.empty((3, h, w), dtype=dt, device=src.device)
    hip.check(hip.lib().refvsr_conv_hr_last_fmt(_ptr(src), h, w, _ptr(blob), act, _ptr(base_lr), bh, bw, _ptr(out), fmt, _stream()), 'conv_hr_last')
    return out


def conv_last_ok(c, h, w):
    return bool(hip.lib().refvsr_conv_last_supported(int(c))) and h * w * c * 2 < 2 ** 31


def conf_alpha_ok(cw):
    return cw.blob24 is not None and cw.cpads == [16] and cw.cout in (24, 48) and not cw.shuffle


def pack_nhwc16(x, cs=None):
    _planar(x)
    c, h, w = x.shape
    cs = cs or (c + 7) // 8 * 8
    out = torch.empty((h, w, cs), dtype=torch.float16, device=x.device)
    hip.check(hip.lib().refvsr_pack_nhwc16(_ptr(x), c, h, w, _ptr(out), cs, _stream()), 'pack_nhwc16')
    return out


def conv1x1_f32(x, w, b, act=0.2):
    """refvsr_conv1x1_f32: fp32 HWC [h][w][cin] -> planar fp32 [16][h][w], 1x1 conv + LeakyReLU (the matching's map64 / map128 block)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous()
    h, w_, cin = x.shape
    assert tuple(w.shape) == (16, cin) and w.dtype == torch.float32 and w.is_contiguous() and tuple(b.shape) == (16,)
    out = torch.empty((16, h, w_), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().refvsr_conv1x1_f32(_ptr(x), cin, h, w_, _ptr(w), _ptr(b), float(act), _ptr(out), _stream()), 'conv1x1_f32')
    return out


def pack_nhwc32(x, cs=None):
    _planar(x)
    c, h, w = x.shape
    cs = cs or (c + 3) // 4 * 4
    out = torch.empty((h, w, cs), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().refvsr_pack_nhwc32(_ptr(x), c, h, w, _ptr(out), cs, _stream()), 'pack_nhwc32')
    return out


def unpack_nhwc16(x, c=None):
    _nhwc(x)
    h, w, cs = x.shape
    c = c or cs
    out = torch.empty((c, h, w), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().refvsr_unpack_nhwc16(_ptr(x), h, w, cs, c, _ptr(out), _stream()), 'unpack_nhwc16')
    return out


def resize(x, out_hw, mode, src_scale=None, mean=None, std=None, chan_mul=None, clamp01=False, nhwc16_out=False):
    """F.interpolate restatement.  src_scale: (sy, sx) source step per output sample; default in/out."""
    _planar(x)
    c, h, w = x.shape
    oh, ow = out_hw
    if src_scale is None:
        src_scale = (float(h) / float(oh), float(w) / float(ow))
    if nhwc16_out:
        out = torch.empty((oh, ow, 8), dtype=torch.float16, device=x.device)
    else:
        out = torch.empty((c, oh, ow), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().refvsr_resize(_ptr(x), c, h, w, _ptr(out), oh, ow, mode, src_scale[0], src_scale[1],
                                      _farr(mean), _farr(std), _farr(chan_mul), int(clamp01), int(nhwc16_out),
                                      8 if nhwc16_out else 0, _stream()), 'resize')
    return out


def bicubic_scale(x, factor, clamp01=True, nhwc16_out=False):
    """F.interpolate(x, scale_factor=factor, mode='bicubic', align_corners=False)[.clamp(0,1)]."""
    c, h, w = x.shape
    oh, ow = int(math.floor(h * factor)), int(math.floor(w * factor))
    s = 1.0 / factor
    return resize(x, (oh, ow), RS_BICUBIC, (s, s), clamp01=clamp01, nhwc16_out=nhwc16_out)


def flow_up2(flow):
    """F.interpolate(flow, scale_factor=2, 'bilinear', align_corners=True) * 2."""
    c, h, w = flow.shape
    return resize(flow, (2 * h, 2 * w), RS_BILINEAR_AC, (0.0, 0.0), chan_mul=[2.0] * c)


def avgpool2(x):
    _planar(x)
    c, h, w = x.shape
    out = torch.empty((c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().refvsr_avgpool2(_ptr(x), c, h, w, _ptr(out), _stream()), 'avgpool2')
    return out


def maxpool2(x):
    _planar(x)
    c, h, w = x.shape
    out = torch.empty((c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().refvsr_maxpool2(_ptr(x), c, h, w, _ptr(out), _stream()), 'maxpool2')
    return out


def max2(a, b):
    assert a.shape == b.shape and a.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous()
    out = torch.empty_like(a)
    hip.check(hip.lib().refvsr_max2(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), 'max2')
    return out


def buffers_equal(pairs):
    """pairs: list of (a, b) float32 tensors of one common shape.  Returns a python list of bools
    (one launch per 32 pairs, one D2H sync)."""
    if not pairs:
        return []
    nbytes = pairs[0][0].numel() * 4
    flags = torch.ones(len(pairs), dtype=torch.int32, device=pairs[0][0].device)
    for s0 in range(0, len(pairs), 32):
        chunk = pairs[s0:s0 + 32]
        for a, b in chunk:
            assert a.shape == b.shape and a.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous()
            assert a.numel() * 4 == nbytes
        pa = (C.c_void_p * len(chunk))(*[a.data_ptr() for a, _ in chunk])
        pb = (C.c_void_p * len(chunk))(*[b.data_ptr() for _, b in chunk])
        hip.check(hip.lib().refvsr_buffers_equal(pa, pb, len(chunk), nbytes, C.c_void_p(flags.data_ptr() + 4 * s0),
                                                 _stream()), 'buffers_equal')
    return [bool(v) for v in flags.cpu().tolist()]


def warp_nhwc16(x, flow):
    _nhwc(x)
    _planar(flow, 2)
    hin, win, cs = x.shape
    hf, wf = flow.shape[1:]
    out = torch.empty((hf, wf, cs), dtype=torch.float16, device=x.device)
    hip.check(hip.lib().refvsr_warp_nhwc16(_ptr(x), hin, win, cs, _ptr(flow), hf, wf, _ptr(out), _stream()), 'warp_nhwc16')
    return out


def warp_nhwc16_up2(x, flow_lr):
    """warp_nhwc16(x, flow_up2(flow_lr)) in one launch (the 2x flow map is evaluated per pixel, never written)."""
    _nhwc(x)
    _planar(flow_lr, 2)
    hin, win, cs = x.shape
    hl, wl = flow_lr.shape[1:]
    out = torch.empty((2 * hl, 2 * wl, cs), dtype=torch.float16, device=x.device)
    hip.check(hip.lib().refvsr_warp_nhwc16_up2(_ptr(x), hin, win, cs, _ptr(flow_lr), hl, wl, _ptr(out), _stream()), 'warp_nhwc16_up2')
    return out


def warp_planar(x, flow):
    _planar(x)
    _planar(flow, 2)
    c, hin, win = x.shape
    hf, wf = flow.shape[1:]
    out = torch.empty((c, hf, wf), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().refvsr_warp_planar(_ptr(x), c, hin, win, _ptr(flow), hf, wf, _ptr(out), _stream()), 'warp_planar')
    return out


def spynet_level_input(ref, supp, flow_prev):
    _planar(ref, 3)
    _planar(supp, 3)
    h, w = ref.shape[1:]
    if flow_prev is not None:
        _planar(flow_prev, 2)
        assert tuple(flow_prev.shape[1:]) == (h // 2, w // 2)
    out8 = torch.empty((h, w, 8), dtype=torch.float16, device=ref.device)
    fup = torch.empty((2, h, w), dtype=torch.float32, device=ref.device)
    hip.check(hip.lib().refvsr_spynet_level_input(_ptr(ref), _ptr(supp), _ptr(flow_prev), h, w, _ptr(out8), _ptr(fup),
                                                  _stream()), 'spynet_level_input')
    return out8, fup


def spynet_level_input_batch(refs, supps, flow_prev):
    """refvsr_spynet_level_input_batch: B independent (ref, supp) pairs of one pyramid level in one launch.  refs / supps:
    lists of B planar fp32 [3,h,w] tensors; flow_prev [B,2,h/2,w/2] or None.  Returns (x [B,h,w,8] fp16, flow_up [B,2,h,w])."""
    B = len(refs)
    assert 1 <= B <= 8 and len(supps) == B
    for t_ in list(refs) + list(supps):
        _planar(t_, 3)
    h, w = refs[0].shape[1:]
    if flow_prev is not None:
        assert flow_prev.is_cuda and flow_prev.dtype == torch.float32 and flow_prev.is_contiguous() and \
            tuple(flow_prev.shape) == (B, 2, h // 2, w // 2)
    out8 = torch.empty((B, h, w, 8), dtype=torch.float16, device=refs[0].device)
    fup = torch.empty((B, 2, h, w), dtype=torch.float32, device=refs[0].device)
    pr = (C.c_void_p * B)(*[t_.data_ptr() for t_ in refs])
    ps = (C.c_void_p * B)(*[t_.data_ptr() for t_ in supps])
    hip.check(hip.lib().refvsr_spynet_level_input_batch(pr, ps, B, _ptr(flow_prev), h, w, _ptr(out8), _ptr(fup), _stream()),
              'spynet_level_input_batch')
    return out8, fup


def match_patches(feat, row_pad, want_lo=False):
    """feat planar [16,h,w] -> (rows fp16 [pad(h*w), KP] zero padded, inv_norm fp32 [h*w][, rows_lo fp16 like rows: the
    low halves of the hi + lo operand split, scaled by 2^11 -- the exact search's second operand])."""
    _planar(feat, 16)
    h, w = feat.shape[1:]
    n = h * w
    # the kernel writes every slot of the n valid rows (incl. the zero pad of each row); only the pad ROWS need clearing
    rows = torch.empty((_round_up(n, row_pad), hip.MATCH_KP), dtype=torch.float16, device=feat.device)
    inv = torch.empty((n,), dtype=torch.float32, device=feat.device)
    lo = torch.empty_like(rows) if want_lo else None
    if rows.shape[0] > n:
        rows[n:].zero_()
        if lo is not None:
            lo[n:].zero_()
    hip.check(hip.lib().refvsr_match_patches(_ptr(feat), h, w, _ptr(rows), _ptr(inv), _ptr(lo), _stream()), 'match_patches')
    return (rows, inv, lo) if want_lo else (rows, inv)


def match_top2(ref_rows, n_ref, lr_rows, n_lr, row_splits=1):
    assert ref_rows.shape[0] % hip.MATCH_ROWCHUNK == 0 and lr_rows.shape[0] % hip.MATCH_COLBLOCK == 0
    assert ref_rows.shape[0] >= n_ref and lr_rows.shape[0] >= n_lr
    ci = torch.empty((n_lr, 2 * row_splits), dtype=torch.int32, device=ref_rows.device)
    cv = torch.empty((n_lr, 2 * row_splits), dtype=torch.float32, device=ref_rows.device)
    hip.check(hip.lib().refvsr_match_top2(_ptr(ref_rows), n_ref, _ptr(lr_rows), n_lr, row_splits, _ptr(ci), _ptr(cv),
                                          _stream()), 'match_top2')
    return ci, cv


# fp16-GEMM scores of rows outside the candidate list are trusted to this margin; columns whose exact maximum does not
# clear the runner-up's fp16 score by it are searched exhaustively at fp32 accuracy (refvsr_match_exact).  The fp16 operand
# rounding perturbs a correlation by ~3e-5 (measured), bounded by 2^-10 = 9.8e-4 in the worst case.
MATCH_EXACT_MARGIN = 2.5e-4


def match_refine(lr_feat, ref_feat, inv_lr, inv_ref, cand, cand_val=None, margin=None, lr_split=None, ref_split=None):
    """Exact re-rank of the candidates; with cand_val / margin / lr_split = (lr_rows, lr_rows_lo) / ref_split = (ref_rows, ref_rows_lo) also the exhaustive
    search of the columns the fp16 GEMM cannot decide.  margin = inf searches EVERY column exhaustively (test aid).
    Returns (conf, idx) or (conf, idx, flagged int32 [1 + n], [0] = count) when flagging is on."""
    _planar(lr_feat, 16)
    _planar(ref_feat, 16)
    h, w = lr_feat.shape[1:]
    hr, wr = ref_feat.shape[1:]
    assert cand.dtype == torch.int32 and cand.shape[0] == h * w and cand.is_contiguous()
    conf = torch.empty((h * w,), dtype=torch.float32, device=lr_feat.device)
    idx = torch.empty((h * w,), dtype=torch.int32, device=lr_feat.device)
    if margin is None:
        hip.check(hip.lib().refvsr_match_refine(_ptr(lr_feat), h, w, _ptr(ref_feat), hr, wr, _ptr(inv_lr), _ptr(inv_ref),
                                                _ptr(cand), None, cand.shape[1], 0.0, None, _ptr(conf), _ptr(idx), _stream()),
                  'match_refine')
        return conf, idx
    assert cand_val is not None and ref_split is not None and lr_split is not None
    assert cand_val.shape == cand.shape and cand_val.is_contiguous()
    (lr_rows, lr_lo), (ref_rows, ref_lo) = lr_split, ref_split
    for r, n_, pad in ((lr_rows, h * w, 1), (lr_lo, h * w, 1), (ref_rows, hr * wr, hip.MATCH_ROWCHUNK), (ref_lo, hr * wr, hip.MATCH_ROWCHUNK)):
        assert r.dtype == torch.float16 and r.is_contiguous() and r.shape[1] == hip.MATCH_KP
        assert r.shape[0] >= n_ and r.shape[0] % pad == 0
    # one zeroed scratch allocation: [flag count + list (int32 1 + n, padded to 8 bytes)] [merge keys uint64 n]
    n = h * w
    fl_words = (n + 2) // 2 * 2
    scratch = torch.zeros(fl_words + 2 * n, dtype=torch.int32, device=lr_feat.device)
    flagged, keys = scratch[:n + 1], scratch[fl_words:]
    hip.check(hip.lib().refvsr_match_refine(_ptr(lr_feat), h, w, _ptr(ref_feat), hr, wr, _ptr(inv_lr), _ptr(inv_ref),
                                            _ptr(cand), _ptr(cand_val), cand.shape[1], float(margin), _ptr(flagged), _ptr(conf),
                                            _ptr(idx), _stream()), 'match_refine')
    hip.check(hip.lib().refvsr_match_exact(_ptr(lr_feat), h, w, _ptr(ref_feat), hr, wr, _ptr(lr_rows), _ptr(lr_lo), _ptr(ref_rows), _ptr(ref_lo),
                                           _ptr(inv_lr), _ptr(inv_ref), _ptr(flagged), _ptr(keys), _ptr(conf), _ptr(idx),
                                           _stream()), 'match_exact')
    return conf, idx, flagged


def match_naive(lr_feat, ref_feat):
    _planar(lr_feat, 16)
    _planar(ref_feat, 16)
    h, w = lr_feat.shape[1:]
    hr, wr = ref_feat.shape[1:]
    conf = torch.empty((h * w,), dtype=torch.float32, device=lr_feat.device)
    idx = torch.empty((h * w,), dtype=torch.int32, device=lr_feat.device)
    hip.check(hip.lib().refvsr_match_naive(_ptr(lr_feat), h, w, _ptr(ref_feat), hr, wr, _ptr(conf), _ptr(idx),
                                           _stream()), 'match_naive')
    return conf, idx


def block_gather_nhwc16(value, idx, gh, gw, s):
    _nhwc(value)
    hv, wv, cs = value.shape
    assert idx.dtype == torch.int32 and idx.numel() == gh * gw and idx.is_contiguous()
    out = torch.empty((gh * s, gw * s, cs), dtype=torch.float16, device=value.device)
    hip.check(hip.lib().refvsr_block_gather_nhwc16(_ptr(value), hv, wv, cs, _ptr(idx), gh, gw, s, _ptr(out), _stream()),
              'block_gather_nhwc16')
    return out


def block_gather_rgb(value, idx, gh, gw, s, planar=False):
    """nhwc16 [gh*s, gw*s, 8] (3 valid channels), or with planar=True an exact planar fp32 copy [3, gh*s, gw*s]."""
    _planar(value, 3)
    hv, wv = value.shape[1:]
    assert idx.dtype == torch.int32 and idx.numel() == gh * gw and idx.is_contiguous()
    if planar:
        out = torch.empty((3, gh * s, gw * s), dtype=torch.float32, device=value.device)
        hip.check(hip.lib().refvsr_block_gather_rgb(_ptr(value), hv, wv, _ptr(idx), gh, gw, s, None, _ptr(out), _stream()),
                  'block_gather_rgb')
        return out
    out = torch.empty((gh * s, gw * s, 8), dtype=torch.float16, device=value.device)
    hip.check(hip.lib().refvsr_block_gather_rgb(_ptr(value), hv, wv, _ptr(idx), gh, gw, s, _ptr(out), None, _stream()),
              'block_gather_rgb')
    return out


def aligned_sample(x, affine, ks):
    _nhwc(x)
    _planar(affine, 3)
    h, w = affine.shape[1:]
    assert tuple(x.shape[:2]) == (h * ks, w * ks)
    out = torch.empty_like(x)
    hip.check(hip.lib().refvsr_aligned_sample(_ptr(x), h, w, ks, x.shape[2], _ptr(affine), _ptr(out), _stream()),
              'aligned_sample')
    return out


# ---- multi-map launches (ABI 11): B maps of one geometry behind one launch per layer ---------------------------------------
# Inputs are LISTS of B tensors (slices of a batched tensor or separately allocated per-frame maps); outputs are ONE tensor with a
# leading batch axis whose slices out[b] are the maps.  Map b of every call == the single-map op on map b, bit for bit.
def _parr(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def multimap_ok(B):
    return 2 <= B <= hip.MAX_MAPS


def _maps(ts, stack):
    """Result of a map-by-map fallback: one [B, ...] tensor like the multi-map launches return (stack: one more copy per map), or
    the list of the B maps as they are (the engine only ever indexes / iterates the result)."""
    return torch.stack(ts, 0) if stack else list(ts)


def conv_b(cw, src0s, src1s=None, act=1.0, muls=None, ress=None, post=1.0, stack=True):
    """refvsr_conv24_batch / refvsr_conv_shuffle2_batch: conv() over B maps (24 output channels, 3x3, or the C = 24 pixel-shuffle
    conv).  Shapes without a multi-map kernel run map by map (same results; stack=False: returned as a list, no copy)."""
    B = len(src0s)
    h, w, c0 = src0s[0].shape
    c1 = src1s[0].shape[2] if src1s is not None else 0
    for l_ in (src0s, src1s, muls, ress):
        if l_ is not None:
            assert len(l_) == B
            for t_ in l_:
                _nhwc(t_)
                assert tuple(t_.shape[:2]) == (h, w)
    dev = src0s[0].device
    if (multimap_ok(B) and cw.shuffle and cw.blob24 is not None and c0 == 24 and src1s is None and muls is None and ress is None and
            0.0 <= act <= 1.0 and post == 1.0 and h * w * cw.cout * 2 < 2 ** 31):
        out = torch.empty((B, 2 * h, 2 * w, c0), dtype=torch.float16, device=dev)
        hip.check(hip.lib().refvsr_conv_shuffle2_batch(_parr(src0s), B, c0, h, w, _ptr(cw.blob24), act, _parr(list(out)), _stream()),
                  'conv_shuffle2_batch')
        return out
    if (multimap_ok(B) and cw.blob24 is not None and not cw.shuffle and cw.cout == 24 and 0.0 <= act <= 1.0 and 0.0 <= post <= 1.0 and
            [c0] + ([c1] if src1s is not None else []) == list(cw.cpads) and hip.lib().refvsr_conv24_supported(c0, c1) and
            (muls is None or muls[0].shape[2] == 24) and (ress is None or ress[0].shape[2] == 24) and h * w * max(24, c0, c1) * 2 < 2 ** 31):
        out = torch.empty((B, h, w, 24), dtype=torch.float16, device=dev)
        hip.check(hip.lib().refvsr_conv24_batch(_parr(src0s), c0, _parr(src1s) if src1s is not None else None, c1, B, h, w, _ptr(cw.blob24),
                                                act, _parr(muls) if muls is not None else None, _parr(ress) if ress is not None else None,
                                                post, _parr(list(out)), _stream()), 'conv24_batch')
        return out
    return _maps([conv(cw, src0s[b], None if src1s is None else src1s[b], act=act, mul=None if muls is None else muls[b],
                       res=None if ress is None else ress[b], post=post) for b in range(B)], stack)


def resblock24_chain_b(chain, xs, act, stack=True):
    """refvsr_resblock24_chain_batch: chain.n fused 24-channel blocks over B maps, one launch per block."""
    B = len(xs)
    for t_ in xs:
        _nhwc(t_)
        assert tuple(t_.shape) == tuple(xs[0].shape) and t_.shape[2] == 24
    if not multimap_ok(B):
        return _maps([resblock24_chain(chain, x, act) for x in xs], stack)
    h, w, _ = xs[0].shape
    dev = xs[0].device
    out = torch.empty((B, h, w, 24), dtype=torch.float16, device=dev)
    s0 = torch.empty_like(out) if chain.n >= 2 else None
    s1 = torch.empty_like(out) if chain.n >= 3 else None
    hip.check(hip.lib().refvsr_resblock24_chain_batch(_parr(xs), B, h, w, chain.n, _ptr(chain.blobs), chain.stride, act, _ptr(s0), _ptr(s1),
                                                      _parr(list(out)), _stream()), 'resblock24_chain_batch')
    return out


def resblock48_chain_b(chain, xs, act, stack=True):
    """refvsr_resblock48_chain_batch (ABI 12): chain.n fused 48-channel blocks over B maps, one launch per block."""
    B = len(xs)
    for t_ in xs:
        _nhwc(t_)
        assert tuple(t_.shape) == tuple(xs[0].shape) and t_.shape[2] == 48
    if not multimap_ok(B):
        return _maps([resblock48_chain(chain, x, act) for x in xs], stack)
    h, w, _ = xs[0].shape
    out = torch.empty((B, h, w, 48), dtype=torch.float16, device=xs[0].device)
    s0 = torch.empty_like(out) if chain.n >= 2 else None
    s1 = torch.empty_like(out) if chain.n >= 3 else None
    hip.check(hip.lib().refvsr_resblock48_chain_batch(_parr(xs), B, h, w, chain.n, _ptr(chain.blobs), chain.stride, act, _ptr(s0), _ptr(s1),
                                                      _parr(list(out)), _stream()), 'resblock48_chain_batch')
    return out


def conf_alpha_b(conf_as, conf_bs, up, w0, b0, cw, slope0=0.2, slope1=0.2, want_max=False, stack=True):
    """refvsr_conf_alpha_batch: conf_alpha() over B pairs of confidence maps (24 output channels)."""
    B = len(conf_as)
    assert len(conf_bs) == B
    if not (multimap_ok(B) and cw.cout == 24):
        r = [conf_alpha(conf_as[b], conf_bs[b], up, w0, b0, cw, slope0, slope1, want_max) for b in range(B)]
        if want_max:
            return _maps([a for a, _ in r], stack), _maps([m for _, m in r], stack)
        return _maps(r, stack)
    for t_ in list(conf_as) + list(conf_bs):
        _planar(t_, 1)
        assert t_.shape == conf_as[0].shape
    assert cw.blob24 is not None and cw.cpads == [16]
    h, w = conf_as[0].shape[1:]
    dev = conf_as[0].device
    alpha = torch.empty((B, up * h, up * w, 24), dtype=torch.float16, device=dev)
    cmax = torch.empty((B, 1, h, w), dtype=torch.float32, device=dev) if want_max else None
    hip.check(hip.lib().refvsr_conf_alpha_batch(_parr(conf_as), _parr(conf_bs), B, h, w, up, _ptr(w0), _ptr(b0), slope0, _ptr(cw.blob24), 24,
                                                slope1, _parr(list(alpha)), _parr(list(cmax)) if want_max else None, _stream()),
              'conf_alpha_batch')
    return (alpha, cmax) if want_max else alpha


def _warp_b(fn, name, xs, flows, out, dims):
    hip.check(fn(_parr(xs), len(xs), *dims[0], _parr(flows), *dims[1], _parr(list(out)), _stream()), name)
    return out


def warp_nhwc16_b(xs, flows, stack=True):
    B = len(xs)
    if not multimap_ok(B):
        return _maps([warp_nhwc16(x, f) for x, f in zip(xs, flows)], stack)
    for x, f in zip(xs, flows):
        _nhwc(x)
        _planar(f, 2)
        assert x.shape == xs[0].shape and f.shape == flows[0].shape
    hin, win, cs = xs[0].shape
    hf, wf = flows[0].shape[1:]
    out = torch.empty((B, hf, wf, cs), dtype=torch.float16, device=xs[0].device)
    return _warp_b(hip.lib().refvsr_warp_nhwc16_batch, 'warp_nhwc16_batch', xs, flows, out, ((hin, win, cs), (hf, wf)))


def warp_nhwc16_up2_b(xs, flows_lr, stack=True):
    B = len(xs)
    if not multimap_ok(B):
        return _maps([warp_nhwc16_up2(x, f) for x, f in zip(xs, flows_lr)], stack)
    for x, f in zip(xs, flows_lr):
        _nhwc(x)
        _planar(f, 2)
        assert x.shape == xs[0].shape and f.shape == flows_lr[0].shape
    hin, win, cs = xs[0].shape
    hl, wl = flows_lr[0].shape[1:]
    out = torch.empty((B, 2 * hl, 2 * wl, cs), dtype=torch.float16, device=xs[0].device)
    return _warp_b(hip.lib().refvsr_warp_nhwc16_up2_batch, 'warp_nhwc16_up2_batch', xs, flows_lr, out, ((hin, win, cs), (hl, wl)))


def warp_planar_b(xs, flows, stack=True):
    B = len(xs)
    if not multimap_ok(B):
        return _maps([warp_planar(x, f) for x, f in zip(xs, flows)], stack)
    for x, f in zip(xs, flows):
        _planar(x)
        _planar(f, 2)
        assert x.shape == xs[0].shape and f.shape == flows[0].shape
    c, hin, win = xs[0].shape
    hf, wf = flows[0].shape[1:]
    out = torch.empty((B, c, hf, wf), dtype=torch.float32, device=xs[0].device)
    return _warp_b(hip.lib().refvsr_warp_planar_batch, 'warp_planar_batch', xs, flows, out, ((c, hin, win), (hf, wf)))


# ---- RefVSR_IR / EDVR-M pieces (csrc/edvr.hip) -------------------------------------------------------------------------
def dcn_sample(x, offset_mask, deform_groups=8):
    """Sampling half of the modulated deformable conv: x nhwc16 [h,w,c], offset_mask planar fp32 [3*dg*9,h,w] (raw
    conv_offset output) -> columns nhwc16 [h,w,9*c] in (tap, channel) order."""
    _nhwc(x)
    h, w, c = x.shape
    _planar(offset_mask, 3 * deform_groups * 9)
    assert tuple(offset_mask.shape[1:]) == (h, w)
    cols = torch.empty((h, w, 9 * c), dtype=torch.float16, device=x.device)
    hip.check(hip.lib().refvsr_dcn_sample(_ptr(x), h, w, c, _ptr(offset_mask), deform_groups, _ptr(cols), _stream()), 'dcn_sample')
    return cols


def tsa_weight(aligned, emb, emb_ref):
    """TSAFusion temporal attention: t aligned maps weighted by sigmoid(<emb_i, emb_ref>), side by side -> [h,w,t*c]."""
    t = len(aligned)
    h, w, c = aligned[0].shape
    for m in list(aligned) + list(emb) + [emb_ref]:
        _nhwc(m)
        assert tuple(m.shape) == (h, w, c)
    out = torch.empty((h, w, t * c), dtype=torch.float16, device=emb_ref.device)
    pa = (C.c_void_p * t)(*[m.data_ptr() for m in aligned])
    pe = (C.c_void_p * t)(*[m.data_ptr() for m in emb])
    hip.check(hip.lib().refvsr_tsa_weight(pa, pe, _ptr(emb_ref), t, c, h * w, _ptr(out), _stream()), 'tsa_weight')
    return out


def pool3s2_pair(x):
    """cat([MaxPool2d(3, 2, 1)(x), AvgPool2d(3, 2, 1)(x)], channel) on an nhwc16 map -> [ho, wo, 2c]."""
    _nhwc(x)
    h, w, c = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty((ho, wo, 2 * c), dtype=torch.float16, device=x.device)
    for off, is_max in ((0, 1), (c, 0)):
        hip.check(hip.lib().refvsr_pool3s2_nhwc16(_ptr(x), h, w, c, _ptr(out), 2 * c, off, is_max, _stream()), 'pool3s2')
    return out


def up2_bilinear_nhwc16(x, mul=1.0):
    _nhwc(x)
    h, w, c = x.shape
    out = torch.empty((2 * h, 2 * w, c), dtype=torch.float16, device=x.device)
    hip.check(hip.lib().refvsr_up2_bilinear_nhwc16(_ptr(x), h, w, c, mul, _ptr(out), _stream()), 'up2_bilinear')
    return out


def tsa_blend(feat, attn, add):
    """feat * sigmoid(attn) * 2 + add."""
    for m in (feat, attn, add):
        _nhwc(m)
        assert m.shape == feat.shape
    out = torch.empty_like(feat)
    hip.check(hip.lib().refvsr_tsa_blend(_ptr(feat), _ptr(attn), _ptr(add), feat.numel(), _ptr(out), _stream()), 'tsa_blend')
    return out
