"""Per-kernel parity on a real MI355X: every C-ABI entry point against the CPU oracle
(oracle/refvsr_oracle.py) on the same seeded inputs.  Integer work (indices, block gathers) is
bit-exact; floating point uses the tolerances written next to each assert (fp16 HWC storage with
fp32 accumulation => ~1e-3 relative per layer; fp32 planar kernels => ~1e-5)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, maxdiff

pytestmark = pytest.mark.gpu
RB24_STORE_DEFAULT = 1            # csrc/resblock24.hip: REFVSR_RB24_STORE_DEFAULT

REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'gpu_ops_report.txt')


def report(name, **kv):
    line = '%-34s ' % name + ' '.join('%s=%.3e' % (k, v) if isinstance(v, float) else '%s=%s' % (k, v) for k, v in kv.items())
    print(line)
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, 'a') as f:
            f.write(line + '\n')
    except OSError:
        pass


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from refvsr_amd import hip
    hip.lib()
    return torch.device('cuda:0')


def nhwc(x, dev, cs=None):
    """planar cpu [C,H,W] -> device nhwc16 via the product's own pack kernel."""
    from refvsr_amd import ops
    return ops.pack_nhwc16(x.contiguous().to(dev), cs)


def planar(x, c=None):
    from refvsr_amd import ops
    return ops.unpack_nhwc16(x, c).cpu()


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# ------------------------------------------------------------------------------------------------
def test_pack_unpack_roundtrip(dev):
    from refvsr_amd import ops
    x = torch.randn(5, 13, 17)
    p = ops.pack_nhwc16(x.to(dev), 8)
    assert p.shape == (13, 17, 8) and p.dtype == torch.float16
    assert torch.equal(p[:, :, 5:].cpu(), torch.zeros(13, 17, 3, dtype=torch.float16))
    assert torch.equal(planar(p, 5), x.half().float())


CONV_CASES = [
    # co, cins, ks, stride, h, w, shuffle
    (24, [24], 3, 1, 19, 45, False),
    (24, [3, 24], 3, 1, 16, 32, False),
    (24, [24, 24], 3, 1, 33, 70, False),
    (96, [24], 3, 1, 18, 40, True),
    (48, [48], 3, 1, 17, 35, False),
    (24, [24], 3, 2, 22, 50, False),
    (32, [3], 5, 1, 20, 36, False),
    (32, [32, 32], 5, 2, 24, 64, False),
    (24, [24, 24], 1, 1, 15, 33, False),
    (32, [8], 7, 1, 18, 30, False),
    (64, [32], 7, 1, 9, 15, False),
    (32, [64], 7, 1, 36, 60, False),
    (16, [32], 7, 1, 12, 20, False),
    (24, [16], 3, 1, 10, 12, False),
    (192, [48], 3, 1, 10, 34, True),
    (48, [48, 48], 3, 1, 12, 40, False),
    # C = 48 at the sizes where the one-workgroup-per-CU variants are selected: 16-wave resident (270 x 480: 510
    # tiles), > 2048 tiles (540 x 960), two-source
    (48, [48], 3, 1, 270, 480, False),
    (48, [48], 3, 1, 540, 960, False),
    (48, [48, 48], 3, 1, 270, 480, False),
    # streamed 7x7 convs at a size with many workgroups (LDS ring, two workgroups per CU)
    (32, [64], 7, 1, 144, 240, False),
    (64, [32], 7, 1, 72, 120, False),
]


@pytest.mark.parametrize('co,cins,ks,stride,h,w,shuffle', CONV_CASES)
def test_conv_mfma_vs_conv2d(dev, co, cins, ks, stride, h, w, shuffle):
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(co * 131 + ks * 7 + sum(cins))
    cin = sum(cins)
    wt = torch.randn(co, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
    b = torch.randn(co, generator=g) * 0.1
    x = torch.randn(1, cin, h, w, generator=g)
    cw = ops.ConvWeights(pack_conv(wt, b, cins, shuffle), dev)
    srcs, o = [], 0
    for c in cins:
        srcs.append(nhwc(x[0, o:o + c], dev))
        o += c
    got = ops.conv(cw, srcs[0], srcs[1] if len(srcs) > 1 else None, stride=stride, act=0.2)
    xh = x.half().float()                      # the kernel sees fp16 activations; weights are hi+lo (~fp32)
    want = F.leaky_relu(F.conv2d(xh, wt, b, stride=stride, padding=ks // 2), 0.2)[0]
    if shuffle:
        want = F.pixel_shuffle(want[None], 2)[0]
    got = planar(got)
    assert got.shape == want.shape
    err = rel(got, want)
    report('conv_mfma co%d cin%s k%d s%d%s' % (co, cins, ks, stride, ' shuf' if shuffle else ''), rel=err)
    assert err < 1e-3          # fp16 output rounding (2^-11 relative) + fp32 accumulation order


@pytest.mark.parametrize('co,ci,h,w,mt', [(64, 32, 9, 15, 1), (32, 64, 18, 30, 1), (32, 64, 36, 60, 2), (64, 32, 72, 120, 1), (2, 16, 9, 15, 1),
                                          (16, 32, 20, 33, 1)])
def test_conv_mfma_streamed_ring(dev, monkeypatch, co, ci, h, w, mt):
    """The streamed (non-resident) 7x7 convs of SPyNet (SPyNet.py:142-202): weight chunks through the LDS-DMA ring with inline-asm
    fragment reads.  16 output channels per workgroup (mt = 1, the packing of the coarse pyramid levels) and the default
    packing give the same map bit for bit, and both match torch; planar fp32 output with residual for the 16 -> 2 flow head."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(co * 7 + ci + h)
    wt = torch.randn(co, ci, 7, 7, generator=g) / (ci * 49) ** 0.5
    b = torch.randn(co, generator=g) * 0.1
    x = torch.randn(1, ci, h, w, generator=g)
    xin = nhwc(x[0], dev)
    xh = x.half().float()
    cw_a = ops.ConvWeights(pack_conv(wt, b, [ci], mt=mt), dev)
    cw_b = ops.ConvWeights(pack_conv(wt, b, [ci]), dev)
    if co == 2:
        res = torch.randn(2, h, w, generator=g)
        got = ops.conv(cw_a, xin, planar_out=True, res_planar=res.to(dev)).cpu()
        want = F.conv2d(xh, wt, b, padding=3)[0] + res
        assert maxdiff(got, want) < 2e-3
        cw_h = ops.ConvWeights(pack_conv(wt, b, [ci], mt=mt, hi_only=True), dev)
        got = ops.conv(cw_h, xin, planar_out=True, res_planar=res.to(dev)).cpu()
        assert maxdiff(got, F.conv2d(xh, wt.half().float(), b, padding=3)[0] + res) < 2e-5       # fp32 output: products exact, sums in fp32
        return
    a = ops.conv(cw_a, xin, act=0.0)
    bb = ops.conv(cw_b, xin, act=0.0)
    want = F.relu(F.conv2d(xh, wt, b, padding=3))[0]
    e = rel(planar(a), want)
    report('conv streamed co%d ci%d %dx%d mt%d' % (co, ci, h, w, mt), rel=e)
    assert e < 1e-3
    assert torch.equal(a, bb), 'mt = %d and default packing differ' % mt
    # weight mode 2 (plain fp16 weights, what Engine.flow runs by default): the same conv with the weights rounded to fp16, to the
    # same bar; against the unrounded weights the difference is the weight rounding itself (2^-11 relative per weight)
    cw_h = ops.ConvWeights(pack_conv(wt, b, [ci], mt=mt, hi_only=True), dev)
    assert cw_h.desc.f32 == 2 and cw_h.wpack.shape[3] == 1 and cw_h.blob24 is None
    c = ops.conv(cw_h, xin, act=0.0)
    e16 = rel(planar(c), F.relu(F.conv2d(xh, wt.half().float(), b, padding=3))[0])
    report('conv streamed fp16 weights co%d ci%d %dx%d mt%d' % (co, ci, h, w, mt), rel=e16, vs_exact_weights=rel(planar(c), want))
    assert e16 < 1e-3 and rel(planar(c), want) < 2e-3


@pytest.mark.parametrize('cins,h,w,act,post,use_mul,use_res', [
    ([24], 19, 45, 0.2, 1.0, False, False), ([24], 270, 480, 1.0, 1.0, False, True), ([24], 61, 130, 0.2, 1.0, True, True),
    ([24], 540, 960, 0.1, 1.0, False, False), ([16], 33, 70, 0.2, 1.0, False, False), ([16], 540, 960, 0.2, 1.0, False, False),
    ([3, 24], 16, 32, 0.1, 1.0, False, False), ([3, 24], 270, 480, 0.1, 1.0, False, False), ([24, 24], 33, 70, 0.2, 1.0, False, False),
    ([24, 24], 540, 960, 0.2, 1.0, False, False), ([24, 24], 7, 5, 1.0, 0.2, False, True), ([24], 8, 32, 0.0, 1.0, False, False),
    ([24], 9, 33, 1.0, 1.0, True, False), ([24], 1080, 1920, 0.1, 1.0, False, False)])
def test_conv24_specialised(dev, cins, h, w, act, post, use_mul, use_res):
    """refvsr_conv24 (csrc/conv24.hip: compile-time K plans, 3-fragment hi + lo rows, bias in the accumulator) for every input
    shape it covers, against torch fp32 on the same fp16 maps and against the runtime-generic refvsr_conv_mfma (same arithmetic
    up to fp32 summation order): interior / border / partial tiles, maps smaller than a tile, all epilogue combinations."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(h * 3 + w + len(cins))
    cin = sum(cins)
    wt = torch.randn(24, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(24, generator=g) * 0.1
    x = torch.randn(1, cin, h, w, generator=g)
    cw = ops.ConvWeights(pack_conv(wt, b, cins), dev)
    assert cw.blob24 is not None
    srcs, o = [], 0
    for c in cins:
        srcs.append(nhwc(x[0, o:o + c], dev))
        o += c
    mul = torch.rand(24, h, w, generator=g) if use_mul else None
    res = torch.randn(24, h, w, generator=g) if use_res else None
    kw = dict(act=act, post=post, mul=nhwc(mul, dev) if use_mul else None, res=nhwc(res, dev) if use_res else None)
    got = ops.conv(cw, srcs[0], srcs[1] if len(srcs) > 1 else None, **kw)
    blob, cw.blob24 = cw.blob24, None                      # the same call through the generic kernel
    gen = ops.conv(cw, srcs[0], srcs[1] if len(srcs) > 1 else None, **kw)
    cw.blob24 = blob
    want = F.leaky_relu(F.conv2d(x.half().float(), wt, b, padding=1), act)[0]
    if use_mul:
        want = want * mul.half().float()
    if use_res:
        want = want + res.half().float()
    want = F.leaky_relu(want, post)
    e, d = rel(planar(got), want), maxdiff(planar(got), planar(gen))
    report('conv24 cin%s %dx%d act%.1f%s%s' % (cins, h, w, act, ' mul' if use_mul else '', ' res' if use_res else ''), rel=e, vs_generic=d)
    assert got.shape == gen.shape == (h, w, 24)
    assert e < 1e-3
    assert d < 4e-3                                        # an fp16 ulp where the fp32 sums round differently



@pytest.mark.parametrize('c,h,w', [(24, 16, 32), (24, 45, 83), (24, 270, 480), (24, 7, 5), (48, 33, 70), (48, 270, 480), (24, 540, 960)])
def test_conv_shuffle2_specialised(dev, c, h, w):
    """refvsr_conv_shuffle2 (PixelShufflePack: C -> 4 C 3x3 conv + F.pixel_shuffle(2), mmedit upsample.py:36-51) on the
    compile-time-specialised kernel vs torch on the same fp16-rounded map and vs the generic kernel's SHUFFLE2 output mode
    (same arithmetic up to fp32 summation order): interior and border tiles, maps smaller than a tile, both channel counts."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(c + h + w)
    wt = torch.randn(4 * c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    b = torch.randn(4 * c, generator=g) * 0.1
    x = torch.randn(c, h, w, generator=g)
    xin = nhwc(x, dev)
    cw = ops.ConvWeights(pack_conv(wt, b, [c], shuffle=True), dev)
    assert cw.blob24 is not None and cw.shuffle
    act = 0.1 if (h + w) % 2 else 1.0                 # upsample2 is followed by a LeakyReLU (RefVSR.py:116), upsample1 is not (:138)
    got = ops.conv(cw, xin, act=act)
    assert got.shape == (2 * h, 2 * w, c)
    blob, cw.blob24 = cw.blob24, None
    generic = ops.conv(cw, xin, act=act)
    cw.blob24 = blob
    want = F.leaky_relu(F.pixel_shuffle(F.conv2d(x.half().float()[None], wt, b, padding=1), 2)[0], act)
    e, eg = rel(planar(got), want), rel(planar(got), planar(generic).float())
    report('conv_shuffle2 c%d %dx%d' % (c, h, w), rel=e, vs_generic=eg)
    assert e < 1e-3 and eg < 1e-3

@pytest.mark.parametrize('cins,h,w,act,post,use_res', [([32], 19, 45, 0.2, 1.0, False), ([32], 540, 960, 1.0, 0.2, True), ([32], 61, 130, 0.2, 1.0, False),
                                                        ([3], 540, 960, 0.2, 1.0, False), ([3], 33, 70, 0.2, 1.0, False), ([32], 7, 5, 1.0, 0.2, True),
                                                        ([32], 270, 480, 1.0, 0.2, True)])
def test_conv32_specialised(dev, cins, h, w, act, post, use_res):
    """refvsr_conv32 (csrc/conv24.hip, COUT = 32: AlignedConv2d's RGB stem and the convs of its 32-channel ResBlocks,
    RefVSR_/alignment.py:18-24 -- conv, LeakyReLU / conv, + x, LeakyReLU) against torch fp32 on the same fp16 maps and against the
    generic kernel: both input shapes (32 channels; 3 -> 8-padded RGB with the one-group K plan), interior / border tiles."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(h * 7 + w + cins[0])
    cin = sum(cins)
    wt = torch.randn(32, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(32, generator=g) * 0.1
    x = torch.randn(1, cin, h, w, generator=g)
    cw = ops.ConvWeights(pack_conv(wt, b, cins), dev)
    assert cw.blob24 is not None and cw.cout == 32
    xin = nhwc(x[0], dev, 8) if cin == 3 else nhwc(x[0], dev)
    res = torch.randn(32, h, w, generator=g) if use_res else None
    kw = dict(act=act, post=post, res=nhwc(res, dev) if use_res else None)
    got = ops.conv(cw, xin, **kw)
    blob, cw.blob24 = cw.blob24, None
    gen = ops.conv(cw, xin, **kw)
    cw.blob24 = blob
    want = F.leaky_relu(F.conv2d(x.half().float(), wt, b, padding=1), act)[0]
    if use_res:
        want = want + res.half().float()
    want = F.leaky_relu(want, post)
    e, d = rel(planar(got), want), maxdiff(planar(got), planar(gen))
    report('conv32 cin%s %dx%d act%.1f post%.1f%s' % (cins, h, w, act, post, ' res' if use_res else ''), rel=e, vs_generic=d)
    assert got.shape == gen.shape == (h, w, 32)
    assert e < 1e-3
    assert d < 4e-3


@pytest.mark.parametrize('cins,h,w,act,use_res', [([48], 19, 45, 0.0, False), ([48], 270, 480, 1.0, True), ([48], 61, 130, 0.2, True),
                                                   ([48], 540, 960, 0.0, False), ([48], 16, 32, 1.0, True), ([48], 17, 33, 0.0, False),
                                                   ([16], 33, 70, 0.2, False), ([16], 270, 480, 0.2, False), ([48], 7, 5, 1.0, True)])
def test_conv48_specialised(dev, cins, h, w, act, use_res):
    """refvsr_conv48 (csrc/conv24.hip, COUT = 48: six fragments per K-step, 16 x 32 tiles on sixteen waves for 48 -> 48; the two
    convs of every residual block of the mid_channels = 48 models) against torch fp32 on the same fp16 maps and against the
    generic kernel: interior / border / partial tiles (16-row tiles: 270 = 16 x 16 + 14), maps smaller than a tile."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(h * 5 + w)
    cin = sum(cins)
    wt = torch.randn(48, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(48, generator=g) * 0.1
    x = torch.randn(1, cin, h, w, generator=g)
    cw = ops.ConvWeights(pack_conv(wt, b, cins), dev)
    assert cw.blob24 is not None and cw.cout == 48
    xin = nhwc(x[0], dev)
    res = torch.randn(48, h, w, generator=g) if use_res else None
    kw = dict(act=act, res=nhwc(res, dev) if use_res else None)
    got = ops.conv(cw, xin, **kw)
    blob, cw.blob24 = cw.blob24, None
    gen = ops.conv(cw, xin, **kw)
    cw.blob24 = blob
    want = F.leaky_relu(F.conv2d(x.half().float(), wt, b, padding=1), act)[0]
    if use_res:
        want = want + res.half().float()
    e, d = rel(planar(got), want), maxdiff(planar(got), planar(gen))
    report('conv48 cin%s %dx%d act%.1f%s' % (cins, h, w, act, ' res' if use_res else ''), rel=e, vs_generic=d)
    assert got.shape == gen.shape == (h, w, 48)
    assert e < 1e-3
    assert d < 4e-3


@pytest.mark.parametrize('cap', [8, 24])
def test_conv_mfma_persistent_tile_walk(dev, cap):
    """Single-chunk convs run on persistent workgroups that walk the pixel tiles (XCD-banded order).  Forcing a tiny
    workgroup count makes every workgroup take many tiles; the result must not depend on the walk."""
    from refvsr_amd import hip, ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(77)
    for (co, cins, ks, h, w, shuffle, f32) in [(24, [24], 3, 37, 70, False, False), (24, [3, 24], 3, 64, 100, False, False),
                                               (96, [24], 3, 41, 33, True, False), (3, [24], 3, 50, 90, False, False),
                                               (16, [64], 1, 45, 77, False, True), (64, [3], 3, 33, 65, False, True)]:
        cin = sum(cins)
        wt = torch.randn(co, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
        b = torch.randn(co, generator=g) * 0.1
        x = torch.randn(cin, h, w, generator=g)
        cw = ops.ConvWeights(pack_conv(wt, b, cins, shuffle, f32=f32), dev)
        srcs, o = [], 0
        for c in cins:
            srcs.append(ops.pack_nhwc32(x[o:o + c].to(dev)) if f32 else nhwc(x[o:o + c], dev))
            o += c
        kw = dict(planar_out=True) if co == 3 else {}
        free = ops.conv(cw, srcs[0], srcs[1] if len(srcs) > 1 else None, act=0.2, **kw).clone()
        hip.lib().refvsr_set_conv_workgroup_cap(cap)
        try:
            walked = ops.conv(cw, srcs[0], srcs[1] if len(srcs) > 1 else None, act=0.2, **kw)
        finally:
            hip.lib().refvsr_set_conv_workgroup_cap(0)
        assert torch.equal(free, walked), (co, cins, ks, h, w)
        xin = x if f32 else x.half().float()
        want = F.leaky_relu(F.conv2d(xin[None], wt, b, padding=ks // 2), 0.2)[0]
        if shuffle:
            want = F.pixel_shuffle(want[None], 2)[0]
        got = walked.cpu() if co == 3 else (walked.permute(2, 0, 1).cpu() if f32 else planar(walked))
        err = rel(got, want)
        report('conv_mfma tile walk cap=%d co%d cin%s k%d' % (cap, co, cins, ks), rel=err)
        assert err < 1e-3


def test_cu_masked_streams_do_not_change_results(dev):
    """CU partitions (ABI 13): a stream bound to CUs [first, first + n) -- refvsr_stream_create_cu_range over
    hipExtStreamCreateWithCUMask -- on which the persistent launchers size their grids for n CUs: the fused block chain (single- and
    four-map launches), a resident and a streamed conv and the matching GEMM give the results of the default stream, bit for bit,
    on two disjoint partitions at once and on a partition of 8 CUs (one per XCD)."""
    from refvsr_amd import hip, ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(31)
    C, n, h, w = 24, 6, 70, 100
    raw = []
    for _ in range(n):
        ws = [torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5 * 0.5 for _ in range(2)]
        raw.append(((ws[0], torch.randn(C, generator=g) * 0.1), (ws[1], torch.randn(C, generator=g) * 0.1)))
    ch = ops.Resblock24Chain(raw, dev)
    xs = [ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev)) for _ in range(4)]
    cw7 = ops.ConvWeights(pack_conv(torch.randn(32, 64, 7, 7, generator=g) * 0.02, torch.zeros(32), [64], False), dev)     # streamed weights
    x64 = ops.pack_nhwc16(torch.randn(64, 40, 56, generator=g).to(dev))
    lr_rows, _ = ops.match_patches(torch.randn(16, 32, 48, generator=g).to(dev), 512)
    ref_rows, _ = ops.match_patches(torch.randn(16, 16, 24, generator=g).to(dev), 256)

    def work():
        return (ops.resblock24_chain(ch, xs[0], 0.0), ops.resblock24_chain_b(ch, xs, 0.2), ops.conv(cw7, x64, act=0.0),
                ops.match_top2(ref_rows, 16 * 24, lr_rows, 32 * 48, 1))
    want = work()
    total = ops.num_cus()
    assert total % 8 == 0 and total >= 64
    half = total // 2 // 8 * 8
    sa, sb, s8 = ops.CuStream(0, half, dev), ops.CuStream(half, total - half, dev), ops.CuStream(8, 8, dev)
    outs = []
    for st in (sa, sb, s8):
        st.wait_stream(torch.cuda.current_stream(dev))
        with ops.on_stream(st):
            outs.append(work())
    torch.cuda.synchronize()
    for got in outs:
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
        assert torch.equal(got[3][0], want[3][0]) and torch.equal(got[3][1], want[3][1])
    lib = hip.lib()
    import ctypes as C_
    assert lib.refvsr_stream_set_cu_budget(C_.c_void_p(sa.cuda_stream), 12) != 0                  # not a multiple of 8
    assert lib.refvsr_stream_set_cu_budget(C_.c_void_p(sa.cuda_stream), total + 8) != 0
    out = C_.c_void_p(0)
    assert lib.refvsr_stream_create_cu_range(4, 8, C_.byref(out)) != 0 and b'multiple-of-8' in lib.refvsr_last_error()


def test_conv_mfma_gather_mode_strided(dev):
    """5x5 stride-4 / stride-8 offset predictors of the HD configs (alignment.py:20): the staged tile cannot fit
    LDS, the kernel switches to gathering B fragments from global memory."""
    from refvsr_amd import hip, ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(77)
    for stride, h, w in ((4, 64, 96), (8, 128, 192), (8, 72, 200)):
        wt = torch.randn(32, 64, 5, 5, generator=g) / (64 * 25) ** 0.5
        b = torch.randn(32, generator=g) * 0.1
        x = torch.randn(1, 64, h, w, generator=g)
        cw = ops.ConvWeights(pack_conv(wt, b, [32, 32]), dev)
        got = planar(ops.conv(cw, nhwc(x[0, :32], dev), nhwc(x[0, 32:], dev), stride=stride, act=0.2))
        want = F.leaky_relu(F.conv2d(x.half().float(), wt, b, stride=stride, padding=2), 0.2)[0]
        assert got.shape == want.shape
        report('conv gather 5x5 s%d %dx%d' % (stride, h, w), rel=rel(got, want))
        assert rel(got, want) < 1e-3


def test_conv_mfma_epilogues(dev):
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(5)
    C, h, w = 24, 21, 37
    wt = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    x, mul, res = (torch.randn(C, h, w, generator=g) for _ in range(3))
    cw = ops.ConvWeights(pack_conv(wt, b, [C]), dev)
    y = F.conv2d(x.half().float()[None], wt, b, padding=1)[0]
    # x + alpha * lrelu(conv)   (RefVSR.py:131)
    got = planar(ops.conv(cw, nhwc(x, dev), act=0.2, mul=nhwc(mul, dev), res=nhwc(res, dev)))
    want = res.half().float() + mul.half().float() * F.leaky_relu(y, 0.2)
    report('conv epilogue mul+res', rel=rel(got, want))
    assert rel(got, want) < 2e-3
    # lrelu(x + conv)   (AlignedConv2d heads, alignment.py:18-20)
    got = planar(ops.conv(cw, nhwc(x, dev), res=nhwc(res, dev), post=0.2))
    want = F.leaky_relu(res.half().float() + y, 0.2)
    assert rel(got, want) < 2e-3
    # relu
    got = planar(ops.conv(cw, nhwc(x, dev), act=0.0))
    assert rel(got, F.relu(y)) < 2e-3
    # planar fp32 output + planar residual + clamp   (conv_last + base, RefVSR.py:118,297)
    w3 = torch.randn(3, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b3 = torch.randn(3, generator=g) * 0.1
    base = torch.rand(3, h, w, generator=g)
    cw3 = ops.ConvWeights(pack_conv(w3, b3, [C]), dev)
    got = ops.conv(cw3, nhwc(x, dev), planar_out=True, res_planar=base.to(dev), clamp=(0.0, 1.0)).cpu()
    want = (F.conv2d(x.half().float()[None], w3, b3, padding=1)[0] + base).clamp(0, 1)
    report('conv planar+res+clamp', abs=maxdiff(got, want))
    assert maxdiff(got, want) < 1e-4
    # planar + constant + clamp(-3,3)   (affine head, alignment.py:47,58)
    got = ops.conv(cw3, nhwc(x, dev), planar_out=True, add_const=1.0, clamp=(-3.0, 3.0)).cpu()
    want = (F.conv2d(x.half().float()[None], w3, b3, padding=1)[0] + 1.0).clamp(-3, 3)
    assert maxdiff(got, want) < 1e-4


@pytest.mark.parametrize('C,h,w,act', [(24, 16, 32, 0.0), (24, 45, 83, 0.2), (24, 270, 480, 0.0), (16, 19, 33, 0.2), (8, 7, 5, 0.1),
                                     (32, 50, 70, 0.2), (24, 540, 960, 0.2)])
def test_resblock_fused(dev, C, h, w, act):
    """Fused conv-act-conv+residual launch vs the same block as two conv launches and vs torch."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    if not ops.resblock_fits(C):
        pytest.skip('fused kernel does not support C=%d' % C)
    g = torch.Generator().manual_seed(C + h)
    w1 = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    w2 = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    x = torch.randn(C, h, w, generator=g)
    c1, c2 = ops.ConvWeights(pack_conv(w1, b1, [C]), dev), ops.ConvWeights(pack_conv(w2, b2, [C]), dev)
    xin = nhwc(x, dev)
    fused = ops.resblock(c1, c2, xin, act=act)
    two = ops.conv(c2, ops.conv(c1, xin, act=act), res=xin)
    xh = x.half().float()
    t = F.leaky_relu(F.conv2d(xh[None], w1, b1, padding=1), act).half().float()       # intermediate is fp16 in both paths
    want = xh + F.conv2d(t, w2, b2, padding=1)[0]
    e_f, e_t = rel(planar(fused), want), rel(planar(two), want)
    same = maxdiff(planar(fused), planar(two))
    report('resblock fused C%d %dx%d act%.1f' % (C, h, w, act), rel_fused=e_f, rel_two=e_t, fused_vs_two=same)
    assert e_f < 1e-3 and e_t < 1e-3
    assert same < 4e-3            # both round the intermediate and the output to fp16; summation order differs
    fused_p = ops.resblock(c1, c2, xin, act=act, post=0.2)
    assert rel(planar(fused_p), F.leaky_relu(want, 0.2)) < 1e-3
    # both workgroup shapes of the lean kernel (8 waves / 4 waves) agree bit for bit
    if ops.hip.lib().refvsr_resblock_lean_fits(C):
        ops.hip.lib().refvsr_set_resblock_waves(4)
        four = ops.resblock(c1, c2, xin, act=act)
        ops.hip.lib().refvsr_set_resblock_waves(8)
        assert torch.equal(four, ops.resblock(c1, c2, xin, act=act)), 'lean 4 waves != 8 waves'


@pytest.mark.parametrize('n', [1, 2, 3, 5, 8])
def test_resblock_chain_call(dev, n):
    """refvsr_resblock_chain (n fused blocks behind one library call, scratch ping-pong) == n calls of the fused block, bit for
    bit; the input map is left untouched; bad buffer aliasing is refused by the library."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    C, h, w = 24, 37, 70
    g = torch.Generator().manual_seed(n)
    pairs = []
    for _ in range(n):
        ws = [torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5 for _ in range(2)]
        bs = [torch.randn(C, generator=g) * 0.1 for _ in range(2)]
        pairs.append(tuple(ops.ConvWeights(pack_conv(ws[i], bs[i], [C]), dev) for i in range(2)))
    x = nhwc(torch.randn(C, h, w, generator=g), dev)
    x0 = x.clone()
    want = x
    for c1, c2 in pairs:
        want = ops.resblock(c1, c2, want, act=0.2)
    assert ops.resblock_chain_ok(C)
    got = ops.resblock_chain(ops.ResblockChain(pairs), x, 0.2)
    assert torch.equal(got, want) and torch.equal(x, x0)
    if n >= 2:
        ch = ops.ResblockChain(pairs)
        lib = ops.hip.lib()
        out, s1 = torch.empty_like(x), torch.empty_like(x)
        rc = lib.refvsr_resblock_chain(x.data_ptr(), C, h, w, n, ch.w1, ch.b1, ch.w2, ch.b2, ch.ksteps, 0.2, 1.0,
                                       x.data_ptr(), s1.data_ptr(), out.data_ptr(), None)  # scratch0 aliases the input
        assert rc != 0 and b'distinct' in lib.refvsr_last_error()


@pytest.mark.parametrize('h,w,act,n', [(16, 32, 0.0, 1), (45, 83, 0.2, 2), (7, 5, 0.0, 3), (270, 480, 0.0, 3), (61, 130, 0.2, 1),
                                     (540, 960, 0.2, 2), (8, 32, 0.0, 1), (9, 33, 0.2, 1)])
def test_resblock24_chain(dev, h, w, act, n):
    """refvsr_resblock24_chain (compile-time-specialised 24-channel kernel, weights as one blob per block) vs torch fp32 on
    the same fp16-rounded maps, vs the runtime-generic lean kernel (same arithmetic up to fp32 summation order), with all three
    workgroup shapes (8 x 32 tiles on 4 / 8 waves, 16 x 32 tiles on 16 waves: bit-identical), interior and border tiles, maps
    smaller than one tile."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    C = 24
    g = torch.Generator().manual_seed(h * 7 + w)
    raw, pairs = [], []
    for _ in range(n):
        ws = [torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5 for _ in range(2)]
        bs = [torch.randn(C, generator=g) * 0.1 for _ in range(2)]
        raw.append(((ws[0], bs[0]), (ws[1], bs[1])))
        pairs.append(tuple(ops.ConvWeights(pack_conv(ws[i], bs[i], [C]), dev) for i in range(2)))
    x = torch.randn(C, h, w, generator=g)
    xin = nhwc(x, dev)
    x0 = xin.clone()
    lib = ops.hip.lib()
    lib.refvsr_set_resblock24_waves(8)
    got = ops.resblock24_chain(ops.Resblock24Chain(pairs, dev), xin, act)      # blobs repacked from the kept raw weights
    got_raw = ops.resblock24_chain(ops.Resblock24Chain(raw, dev), xin, act)
    assert torch.equal(got, got_raw) and torch.equal(xin, x0)
    for waves in (4, 16, 0):                    # 8 x 32 tiles on 4 waves, 16 x 32 tiles on 16 waves, the size-dependent default
        lib.refvsr_set_resblock24_waves(waves)
        other = ops.resblock24_chain(ops.Resblock24Chain(raw, dev), xin, act)
        lib.refvsr_set_resblock24_waves(0)
        assert torch.equal(got, other), 'workgroup shape %d differs from 8 waves' % waves
    want, lean = x.half().float(), xin
    for ((w1, b1), (w2, b2)), (c1, c2) in zip(raw, pairs):
        t = F.leaky_relu(F.conv2d(want[None], w1, b1, padding=1), act).half().float()
        want = (want + F.conv2d(t, w2, b2, padding=1)[0]).half().float()
        lean = ops.resblock(c1, c2, lean, act=act)
    e, d = rel(planar(got), want), maxdiff(planar(got), planar(lean))
    report('resblock24 %dx%d act%.1f n%d' % (h, w, act, n), rel=e, vs_lean=d)
    assert e < 1e-3 * n
    assert d < 4e-3 * n            # fp16 ulps where the fp32 sums round differently (|x| < 8: ulp 4e-3 .. 8e-3)


def test_conv_mfma_f32_mode(dev):
    """Exact-fp32 MFMA mode (v_mfma_f32_16x16x4_f32) used for the VGG feature extractor."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(21)
    for (co, ci, k, h, w) in [(64, 3, 3, 20, 28), (64, 64, 3, 35, 50), (16, 64, 1, 20, 28), (64, 64, 3, 70, 130)]:
        wt = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
        b = torch.randn(co, generator=g) * 0.1
        x = torch.randn(ci, h, w, generator=g)
        cw = ops.ConvWeights(pack_conv(wt, b, [ci], f32=True), dev)
        xin = ops.pack_nhwc32(x.to(dev))
        want = F.relu(F.conv2d(x[None], wt, b, padding=k // 2))[0]
        got = ops.conv(cw, xin, act=0.0)                      # fp32 HWC out
        assert got.dtype == torch.float32 and got.shape == (h, w, co)
        e1 = maxdiff(got.permute(2, 0, 1).cpu(), want)
        got_p = ops.conv(cw, xin, act=0.0, planar_out=True).cpu()
        report('conv_mfma f32 co%d ci%d k%d' % (co, ci, k), abs=e1, abs_planar=maxdiff(got_p, want))
        assert e1 < 2e-5 and maxdiff(got_p, want) < 2e-5      # fp32 FMA chain vs MKL-DNN summation order


def test_conv_direct_f32(dev):
    from refvsr_amd import ops
    g = torch.Generator().manual_seed(9)
    for (co, ci, k, s, h, w) in [(64, 3, 3, 1, 20, 28), (64, 64, 3, 1, 20, 28), (16, 64, 1, 1, 20, 28), (3, 3, 1, 1, 9, 11),
                                 (16, 2, 3, 1, 17, 19), (128, 64, 3, 1, 10, 14), (24, 5, 3, 2, 18, 22)]:
        wt = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
        b = torch.randn(co, generator=g) * 0.1
        x = torch.randn(ci, h, w, generator=g)
        got = ops.conv_direct(x.to(dev), wt.to(dev), b.to(dev), stride=s, act=0.2).cpu()
        want = F.leaky_relu(F.conv2d(x[None], wt, b, stride=s, padding=k // 2), 0.2)[0]
        report('conv_direct co%d ci%d k%d s%d' % (co, ci, k, s), abs=maxdiff(got, want))
        assert maxdiff(got, want) < 2e-5           # fp32 FMA chain vs MKL-DNN blocking
        if co % 8 == 0:
            got16 = planar(ops.conv_direct(x.to(dev), wt.to(dev), b.to(dev), stride=s, act=0.2, nhwc16_out=True))
            assert rel(got16, want) < 1e-3


def test_resize_modes_vs_golden_and_oracle(dev):
    from refvsr_amd import ops
    from oracle import refvsr_oracle as orc
    from refvsr_amd.weights import VGG_MEAN, VGG_STD
    g = load_golden('op_resize')
    img, fl = g['img'][0].to(dev), g['flow'][0].to(dev)
    errs = {}
    errs['bicubic_half'] = maxdiff(ops.bicubic_scale(img, 0.5, clamp01=False).cpu(), g['bicubic_half'][0])
    errs['bicubic_x2'] = maxdiff(ops.bicubic_scale(img, 2, clamp01=False).cpu(), g['bicubic_x2'][0])
    errs['bicubic_x4'] = maxdiff(ops.bicubic_scale(img, 4, clamp01=False).cpu(), g['bicubic_x4'][0])
    errs['bicubic_x4_clamp'] = maxdiff(ops.bicubic_scale(img, 4, clamp01=True).cpu(), g['bicubic_x4'][0].clamp(0, 1))
    errs['flow_up2'] = maxdiff(ops.flow_up2(fl).cpu(), g['flow_up2'][0]) / 6.0          # values up to ~6
    errs['bilinear_up'] = maxdiff(ops.resize(img, (32, 32), ops.RS_BILINEAR).cpu(), g['bilinear_32x32'][0])
    errs['bilinear_back'] = maxdiff(ops.resize(g['bilinear_32x32'][0].to(dev), (18, 26), ops.RS_BILINEAR).cpu(), g['bilinear_back'][0])
    errs['nearest'] = maxdiff(ops.resize(img, (9, 13), ops.RS_NEAREST, (2.0, 2.0)).cpu(), g['nearest_half'][0])
    # larger, odd geometry + fused normalisation / per-channel gain / nhwc16 output
    x = torch.rand(3, 54, 100)
    got = ops.resize(x.to(dev), (64, 128), ops.RS_BILINEAR, mean=VGG_MEAN, std=VGG_STD).cpu()
    want = (orc.resize(x[None], (64, 128), 'bilinear')[0] - torch.tensor(VGG_MEAN).view(3, 1, 1)) / torch.tensor(VGG_STD).view(3, 1, 1)
    errs['bilinear_norm'] = maxdiff(got, want) / 3.0
    f2 = torch.randn(2, 64, 128)
    got = ops.resize(f2.to(dev), (54, 100), ops.RS_BILINEAR, chan_mul=[100 / 128.0, 54 / 64.0]).cpu()
    want = orc.resize(f2[None], (54, 100), 'bilinear')[0] * torch.tensor([100 / 128.0, 54 / 64.0]).view(2, 1, 1)
    errs['bilinear_mul'] = maxdiff(got, want) / 3.0
    got = planar(ops.bicubic_scale(x.to(dev), 2, clamp01=False, nhwc16_out=True), 3)
    errs['bicubic_nhwc16'] = maxdiff(got, orc.bicubic_scale(x[None], 2, False)[0]) / 50.0    # fp16 store: 1e-3 allowed
    report('resize', **errs)
    assert errs.pop('nearest') == 0.0
    for k, v in errs.items():
        assert v < 2e-5, (k, v)                     # fp32 interpolation arithmetic


def test_pools_and_max(dev):
    from refvsr_amd import ops
    from oracle import refvsr_oracle as orc
    x = torch.randn(5, 18, 26)
    assert maxdiff(ops.avgpool2(x.to(dev)).cpu(), orc.avg_pool2(x[None])[0]) < 1e-6
    assert maxdiff(ops.maxpool2(x.to(dev)).cpu(), orc.max_pool2(x[None])[0]) == 0.0
    y = torch.randn(5, 18, 26)
    assert torch.equal(ops.max2(x.to(dev), y.to(dev)).cpu(), torch.maximum(x, y))


def test_warp_vs_golden(dev):
    from refvsr_amd import ops
    g = load_golden('op_warp')
    x, fl, fl2 = g['x'][0], g['flow'][0], g['flow2'][0]
    got = ops.warp_planar(x.to(dev), fl.to(dev)).cpu()
    report('warp_planar', abs=maxdiff(got, g['warp'][0]))
    assert maxdiff(got, g['warp'][0]) < 2e-5
    assert maxdiff(ops.warp_planar(x.to(dev), fl2.to(dev)).cpu(), g['warp2'][0]) < 2e-5     # LR input, 2x flow
    xh = x.half().float()
    from oracle import refvsr_oracle as orc
    for f_, name in ((fl, 'lr'), (fl2, '2x')):
        got = planar(ops.warp_nhwc16(nhwc(x, dev), f_.to(dev)), 5)
        want = orc.warp(xh[None], f_[None])[0]
        report('warp_nhwc16 ' + name, abs=maxdiff(got, want))
        assert maxdiff(got, want) < 1.5e-3      # fp16 output rounding of O(1) values
    # zero flow is NOT the identity: column 0 samples at -0.5
    z = torch.zeros(2, 18, 26)
    got = ops.warp_planar(x.to(dev), z.to(dev)).cpu()
    assert maxdiff(got, orc.warp(x[None], z[None])[0]) < 2e-5
    assert maxdiff(got, x) > 1e-2


def test_spynet_level_input(dev):
    from refvsr_amd import ops
    from oracle import refvsr_oracle as orc
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(3, 18, 30, generator=g), torch.randn(3, 18, 30, generator=g)
    fp = torch.randn(2, 9, 15, generator=g) * 3
    out8, fup = ops.spynet_level_input(a.to(dev), b.to(dev), fp.to(dev))
    want_up = orc.flow_up2(fp[None])
    assert maxdiff(fup.cpu(), want_up[0]) < 2e-5
    want = torch.cat([a, orc.flow_warp_border(b[None], want_up)[0], want_up[0]], 0)
    got = planar(out8)
    report('spynet_level_input', abs=maxdiff(got, want))
    assert maxdiff(got, want) < 8e-3            # fp16 store of values up to ~8
    out8, fup = ops.spynet_level_input(a.to(dev), b.to(dev), None)
    assert float(fup.abs().max()) == 0.0
    # zero flow: the normalise/unnormalise round trip of flow_warp leaves ~1e-5 of interpolation
    assert maxdiff(planar(out8)[3:6], orc.flow_warp_border(b[None], torch.zeros(1, 2, 18, 30))[0]) < 4e-3
    gw = load_golden('op_warp')                 # mmedit flow_warp fixture (produced by the reference) through the kernel
    x3, fl = gw['x'][0, :3], gw['flow'][0]
    out8, _ = ops.spynet_level_input(x3.to(dev), x3.to(dev), (fl[:, ::2, ::2] * 0).contiguous().to(dev))
    assert maxdiff(planar(out8)[3:6], x3) < 2e-3


def test_match_patches(dev):
    from refvsr_amd import ops
    from oracle import refvsr_oracle as orc
    f = torch.randn(16, 14, 18)
    rows, inv = ops.match_patches(f.to(dev), 256)
    p = orc.patches3x3(f[None])[0].t()                          # [L,144]
    n = p.norm(dim=1, keepdim=True).clamp_min(1e-12)
    assert rows.shape == (256, 152) and float(rows[14 * 18:].abs().max()) == 0 and float(rows[:, 144:].abs().max()) == 0
    assert maxdiff(rows[:14 * 18, :144].float().cpu(), p / n) < 5e-4     # fp16 rounding of |v| <= 1
    assert maxdiff(inv.cpu() * n[:, 0], torch.ones(14 * 18)) < 1e-5


def _check_match(conf, idx, lr_f, ref_f, tag):
    """conf/idx from the HIP path vs the oracle GEMM: the chosen index must achieve the oracle's column
    maximum to within fp32 rounding (ties/near-ties may legitimately pick another index)."""
    from oracle import refvsr_oracle as orc
    lp = orc.patches3x3(lr_f[None])
    rp = orc.patches3x3(ref_f[None]).permute(0, 2, 1)
    rp = rp / rp.norm(dim=2, keepdim=True).clamp_min(1e-12)
    lp = lp / lp.norm(dim=1, keepdim=True).clamp_min(1e-12)
    corr = torch.bmm(rp, lp)[0]                                  # [n_ref, n_lr]
    val, ix = corr.max(0)
    conf, idx = conf.cpu(), idx.cpu().long()
    achieved = corr.gather(0, idx[None])[0]
    mism = int((idx != ix).sum())
    report('match ' + tag, conf_err=maxdiff(conf, val), idx_mismatch=mism, n=idx.numel(),
           worst_gap=float((val - achieved).max()))
    assert maxdiff(conf, val) < 5e-6
    assert float((val - achieved).max()) < 5e-6
    assert mism <= max(2, idx.numel() // 2000)
    return mism


def test_match_fused_vs_oracle(dev):
    from refvsr_amd import ops
    g = torch.Generator().manual_seed(11)
    for (h, w) in [(20, 28), (34, 50), (64, 96)]:
        base = F.interpolate(torch.randn(1, 16, h // 4 + 2, w // 4 + 2, generator=g), size=(h, w), mode='bilinear')[0]
        lr_f = base + 0.2 * torch.randn(16, h, w, generator=g)
        ref_f = F.avg_pool2d(base[None], 2)[0] + 0.2 * torch.randn(16, h // 2, w // 2, generator=g)
        lr_rows, inv_lr = ops.match_patches(lr_f.to(dev), 512)
        ref_rows, inv_ref = ops.match_patches(ref_f.to(dev), 256)
        n_ref = ref_f.shape[1] * ref_f.shape[2]
        for splits in (1, 2):
            if splits > (n_ref + 255) // 256:
                continue
            cand, cval = ops.match_top2(ref_rows, n_ref, lr_rows, h * w, splits)
            assert int(cand.min()) >= 0 and int(cand.max()) < n_ref
            conf, idx = ops.match_refine(lr_f.to(dev), ref_f.to(dev), inv_lr, inv_ref, cand)
            _check_match(conf, idx, lr_f, ref_f, '%dx%d splits=%d' % (h, w, splits))
        conf, idx = ops.match_naive(lr_f.to(dev), ref_f.to(dev))
        _check_match(conf, idx, lr_f, ref_f, '%dx%d naive' % (h, w))


def test_match_ties_pick_first_index(dev):
    """Constant features: every correlation ties exactly; torch.max returns index 0."""
    from refvsr_amd import ops
    lr_f = torch.ones(16, 12, 16)
    ref_f = torch.ones(16, 6, 8)
    lr_rows, inv_lr = ops.match_patches(lr_f.to(dev), 512)
    ref_rows, inv_ref = ops.match_patches(ref_f.to(dev), 256)
    cand, _ = ops.match_top2(ref_rows, 48, lr_rows, 192, 1)
    conf, idx = ops.match_refine(lr_f.to(dev), ref_f.to(dev), inv_lr, inv_ref, cand)
    assert int(idx.abs().max()) == 0
    assert maxdiff(conf.cpu(), torch.ones(192)) < 1e-5


def test_match_patches_split_rows(dev):
    """hi + lo operand split of the normalised patch rows: hi + lo * 2^-11 reproduces the fp32 normalised patch to 2^-22."""
    from refvsr_amd import ops
    from oracle import refvsr_oracle as orc
    f = torch.randn(16, 14, 18)
    rows, inv, lo = ops.match_patches(f.to(dev), 256, want_lo=True)
    n = 14 * 18
    want = F.normalize(orc.patches3x3(f[None])[0].t().contiguous().double(), dim=1)          # [n, 144]
    got = rows[:n, :144].cpu().double() + lo[:n, :144].cpu().double() / 2048.0
    report('match split rows', err=float((got - want).abs().max()), hi_only=float((rows[:n, :144].cpu().double() - want).abs().max()))
    assert float((got - want).abs().max()) < 2e-7
    assert float(rows[n:].abs().max()) == 0 and float(lo[:, 144:].abs().max()) == 0 and float(rows[:, 144:].abs().max()) == 0


def test_match_exact_search(dev):
    """refvsr_match_exact (exhaustive split-fp16 MFMA search of flagged columns): with margin = inf EVERY column is flagged,
    the result must then be the exact arg-max and -- where the default margin flags nothing -- identical to the top-2
    path; conf values of both paths come from the same fp32 expression."""
    from refvsr_amd import ops
    g = torch.Generator().manual_seed(13)
    for (h, w) in [(20, 28), (34, 50), (64, 96)]:
        base = F.interpolate(torch.randn(1, 16, h // 4 + 2, w // 4 + 2, generator=g), size=(h, w), mode='bilinear')[0]
        lr_f = base + 0.2 * torch.randn(16, h, w, generator=g)
        ref_f = F.avg_pool2d(base[None], 2)[0] + 0.2 * torch.randn(16, h // 2, w // 2, generator=g)
        lr_rows, inv_lr, lr_lo = ops.match_patches(lr_f.to(dev), 512, want_lo=True)
        ref_rows, inv_ref, ref_lo = ops.match_patches(ref_f.to(dev), 256, want_lo=True)
        n_ref = ref_f.shape[1] * ref_f.shape[2]
        cand, cval = ops.match_top2(ref_rows, n_ref, lr_rows, h * w, 1)
        split = ((lr_rows, lr_lo), (ref_rows, ref_lo))
        conf_a, idx_a, fl_a = ops.match_refine(lr_f.to(dev), ref_f.to(dev), inv_lr, inv_ref, cand, cval, float('inf'), *split)
        assert int(fl_a[0]) == h * w
        _check_match(conf_a, idx_a, lr_f, ref_f, '%dx%d exhaustive exact' % (h, w))
        conf_d, idx_d, fl_d = ops.match_refine(lr_f.to(dev), ref_f.to(dev), inv_lr, inv_ref, cand, cval, ops.MATCH_EXACT_MARGIN, *split)
        conf_t, idx_t = ops.match_refine(lr_f.to(dev), ref_f.to(dev), inv_lr, inv_ref, cand)
        report('match exact %dx%d' % (h, w), flagged_default=int(fl_d[0]), idx_diff_all_vs_top2=int((idx_a != idx_t).sum()),
               conf_bitwise_equal=float(torch.equal(conf_a, conf_t)), conf_diff=maxdiff(conf_a.cpu(), conf_t.cpu()))
        # the search winner is re-evaluated with the re-rank's fp32 expression and only replaces a candidate it beats:
        # same row => same number, different row => a larger (or equal, smaller index) value
        same = idx_a == idx_t
        assert torch.equal(conf_a[same], conf_t[same])
        assert bool((conf_a[~same] >= conf_t[~same]).all())
        assert torch.equal(idx_d, idx_a) and torch.equal(conf_d, conf_a)      # default margin: same final answer as exhaustive


def test_match_near_ties_flat_regions(dev, small_cfg, small_sd):
    """8-bit frames with large flat areas (+-1 LSB noise, soft gradients, repeated texture): many reference patches are
    nearly equally similar, the fp16 GEMM's top-2 can miss the true maximum.  Whole FeatureMatching path vs the oracle:
    per column the correlation reached by the HIP index must be >= the oracle's maximum - 1e-6, and an exact (all
    constant) region must pick the first index like torch.max."""
    from oracle import refvsr_oracle as orc
    from refvsr_amd.engine import Engine, FrameCtx, Weights
    rs = np.random.RandomState(5)
    h, w = 96, 128
    img = np.full((3, h, w), 0.5, np.float32)
    img[:, :, 64:] = 0.25                                               # two flat halves
    img[:, 20:50, 10:60] += (rs.randint(-1, 2, (3, 30, 50)) / 255.0)  # +-1 LSB noise patch
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img[:, 60:, :] = 0.3 + 0.2 * (xx[60:] / w)[None]                    # soft gradient (quantised below)
    img[:, 70:90, 70:120] = 0.5 + 0.25 * np.sign(np.sin(xx[70:90, 70:120] * 0.8))[None]   # repeated stripes
    lr = torch.from_numpy(np.round(img * 255.0) / 255.0)
    ref = torch.roll(lr, shifts=(3, -5), dims=(1, 2)).contiguous()
    eng = Engine(small_cfg, Weights(small_cfg, small_sd, dev))
    conf, idx, _ = eng.feature_match(FrameCtx(lr.to(dev), ref.to(dev)))
    ref_p, lr_p, _ = orc.match_features(lr[None], ref[None], small_sd, False)
    corr = torch.bmm(ref_p.double(), lr_p.double())[0]                  # [n_ref, n_lr] in fp64
    val, ix = corr.max(0)
    idx = idx.cpu().long()
    achieved = corr.gather(0, idx[None])[0]
    gap = float((val - achieved).max())
    # how often would the top-2 list alone have missed? (diagnostic: the exhaustive search is what closes the gap)
    eng2 = Engine(small_cfg, Weights(small_cfg, small_sd, dev))
    eng2.match_margin = 0.0
    _, idx2, _ = eng2.feature_match(FrameCtx(lr.to(dev), ref.to(dev)))
    gap2 = float((val - corr.gather(0, idx2.cpu().long()[None])[0]).max())
    report('match near-ties', worst_gap=gap, worst_gap_top2_only=gap2, idx_mismatch=int((idx != ix).sum()), n=idx.numel(),
           conf_err=maxdiff(conf.cpu().view(-1), val.float()))
    assert gap < 1e-6
    assert maxdiff(conf.cpu().view(-1), val.float()) < 2e-6
    # exact ties (identical patches): smallest index, like torch.max
    o_conf, o_idx = orc.feature_match(lr[None], ref[None], small_sd, False)
    flat = (o_conf.view(-1) > 1.0 - 1e-6)
    mism = int((idx[flat] != o_idx[0][flat]).sum())
    report('match near-ties exact-tie columns', n=int(flat.sum()), mismatch=mism)
    assert int(flat.sum()) > 1000 and mism == 0      # exact ties pick the first index, every one of them (torch.max semantics)


def test_feature_match_golden(dev, small_cfg, small_sd):
    """Whole FeatureMatching.forward against the fixture produced by the reference."""
    from refvsr_amd.engine import Engine, FrameCtx, Weights
    g = load_golden('op_match')
    eng = Engine(small_cfg, Weights(small_cfg, small_sd, dev))
    conf, idx, _ = eng.feature_match(FrameCtx(g['lr'][0].to(dev), g['ref'][0].to(dev)))
    mism = int((idx.cpu().long() != g['idx'][0]).sum())
    report('feature_match golden', conf_err=maxdiff(conf.cpu(), g['conf'][0]), idx_mismatch=mism)
    assert maxdiff(conf.cpu(), g['conf'][0]) < 2e-5
    assert mism <= 1


def test_block_gather_bit_exact(dev):
    from refvsr_amd import ops
    g = load_golden('op_aa')
    idx = g['idx'][0].to(torch.int32).to(dev)
    v, vd, rf = g['value'][0], g['value_down'][0], g['ref'][0]
    assert torch.equal(planar(ops.block_gather_nhwc16(nhwc(vd, dev), idx, 20, 28, 1)), g['aa1'][0].half().float())
    assert torch.equal(planar(ops.block_gather_nhwc16(nhwc(v, dev), idx, 20, 28, 2)), g['aa2_fm'][0].half().float())
    assert torch.equal(planar(ops.block_gather_rgb(rf.to(dev), idx, 20, 28, 2), 3), g['aa2_rgb'][0].half().float())


def test_aligned_sample(dev):
    from refvsr_amd import ops
    from oracle import refvsr_oracle as orc
    g = load_golden('op_sampler')
    x, aff = g['x'][0], g['affine'][0]
    x8 = torch.cat([x, torch.zeros(5, 12, 16)], 0)
    got = planar(ops.aligned_sample(nhwc(x8, dev), aff.to(dev), 2), 3)
    report('aligned_sample golden', abs=maxdiff(got, g['out'][0]))
    assert maxdiff(got, g['out'][0]) < 1.5e-3         # fp16 in/out of O(1) values
    gen = torch.Generator().manual_seed(2)
    for ks in (2, 4):
        h, w = 7, 9
        xx = torch.randn(16, h * ks, w * ks, generator=gen)
        af = torch.stack([torch.rand(h, w, generator=gen) * 6 - 3, torch.rand(h, w, generator=gen) * 6 - 3,
                          torch.rand(h, w, generator=gen) * 6 - 3])
        got = planar(ops.aligned_sample(nhwc(xx, dev), af.to(dev), ks))
        want = orc.aligned_sample(xx.half().float()[None], af[None], ks)[0]
        report('aligned_sample ks%d' % ks, abs=maxdiff(got, want))
        assert maxdiff(got, want) < 4e-3
        ident = planar(ops.aligned_sample(nhwc(xx, dev), torch.ones(3, h, w).to(dev), ks))
        assert maxdiff(ident, xx.half().float()) < 2e-3


def test_spynet_flow_golden(dev, small_cfg, small_sd):
    from refvsr_amd.engine import Engine, FrameCtx, Weights
    g = load_golden('op_spynet')
    eng = Engine(small_cfg, Weights(small_cfg, small_sd, dev))
    a, b = g['a'][0].to(dev), g['b'][0].to(dev)
    fl = eng.flow(FrameCtx(a, a), FrameCtx(b, b)).cpu()
    report('spynet golden', abs=maxdiff(fl, g['flow'][0]), flow_mag=float(g['flow'].abs().max()))
    # pixels; measured 3.9e-4 (fp16 conv stack, plain fp16 weights in the streamed 7x7 convs -- round 3), bar = 2x
    assert maxdiff(fl, g['flow'][0]) < 8e-4
    import copy
    cfg2 = copy.deepcopy(small_cfg)
    cfg2.spynet_hi_lo = True                         # hi + lo weights everywhere: measured 1.9e-4, bar = 2x
    eng2 = Engine(cfg2, Weights(cfg2, small_sd, dev))
    fl2 = eng2.flow(FrameCtx(a, a), FrameCtx(b, b)).cpu()
    report('spynet golden hi+lo', abs=maxdiff(fl2, g['flow'][0]))
    assert maxdiff(fl2, g['flow'][0]) < 4e-4


def test_conv_stacks_golden(dev, small_cfg, small_sd):
    from refvsr_amd import ops
    from refvsr_amd.engine import Engine, Weights
    g = load_golden('op_convs')
    eng = Engine(small_cfg, Weights(small_cfg, small_sd, dev))
    feat, img = g['feat'][0], g['img'][0]
    got = planar(eng.res_list(nhwc(feat, dev), 'feat_decoder2', 4))
    report('res_list golden', abs=maxdiff(got, g['res_list'][0]), mag=float(g['res_list'].abs().max()))
    assert maxdiff(got, g['res_list'][0]) < 1.3e-3     # measured 6.4e-4, bar = 2x
    got = planar(eng.resblocks(nhwc(img, dev, 8), nhwc(feat, dev), 'backward_resblocks'))
    report('resblocks golden', abs=maxdiff(got, g['resblocks'][0]), mag=float(g['resblocks'].abs().max()))
    assert maxdiff(got, g['resblocks'][0]) < 3.3e-3   # 49 fp16 layers; measured 1.6e-3, bar = 2x
    got = planar(ops.conv(eng.cw('upsample1.upsample_conv'), nhwc(feat, dev)))
    assert maxdiff(got, g['pixel_shuffle'][0]) < 3e-3


def test_compute_up_golden(dev, small_cfg, small_sd):
    from refvsr_amd.engine import Engine, Weights
    g = load_golden('op_compute_up')
    eng = Engine(small_cfg, Weights(small_cfg, small_sd, dev))
    # base is an input of the reference's compute_up; feed lr such that bicubic x4 is not needed: compare pre-base
    from oracle import refvsr_oracle as orc
    o = orc.OracleNetwork(small_cfg, small_sd)
    lr = torch.rand(1, 3, 6, 8)
    base = orc.bicubic_scale(lr, 4, True)
    want = o._compute_up(g['bw'], g['fw'], g['conf_bw'], g['conf_fw'], base).clamp(0, 1)[0]
    got = eng.compute_up(nhwc(g['bw'][0], dev), nhwc(g['fw'][0], dev), g['conf_bw'][0].to(dev), g['conf_fw'][0].to(dev),
                         lr[0].to(dev)).cpu()
    report('compute_up', abs=maxdiff(got, want))
    assert maxdiff(got, want) < 1.1e-3               # measured 5.3e-4, bar = 2x


# ------------------------------------------------------------------------------------------------
# RefVSR_IR / EDVR-M pieces (csrc/edvr.hip) against the IR oracle's restatements
# ------------------------------------------------------------------------------------------------
def test_dcn_modulated_deformable_conv(dev):
    """conv_offset -> dcn_sample -> 1x1 contraction == ModulatedDCNPack of the oracle (edvr_net.py:49-56), incl. samples
    that leave the map (zero outside, partial corners) and large offsets."""
    from oracle import refvsr_ir_oracle as iro
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(41)
    M, h, w = 64, 23, 31
    x = torch.randn(1, M, h, w, generator=g)
    extra = torch.randn(1, M, h, w, generator=g)
    W = {'d.weight': torch.randn(M, M, 3, 3, generator=g) / 24.0, 'd.bias': torch.randn(M, generator=g) * 0.1,
         'd.conv_offset.weight': torch.randn(216, M, 3, 3, generator=g) / 12.0, 'd.conv_offset.bias': torch.randn(216, generator=g)}
    want = iro.dcn_pack(x, extra, W, 'd')[0]
    co = ops.ConvWeights(pack_conv(W['d.conv_offset.weight'], W['d.conv_offset.bias'], [M]), dev)
    cd = ops.ConvWeights(pack_conv(W['d.weight'].permute(0, 2, 3, 1).reshape(M, 9 * M, 1, 1), W['d.bias'], [9 * M]), dev)
    om = ops.conv(co, nhwc(extra[0], dev), planar_out=True)
    got = planar(ops.conv(cd, ops.dcn_sample(nhwc(x[0], dev), om, 8)))
    # offsets reach several pixels here: their fp16-operand error (1e-3 px) times the map's gradient dominates
    report('dcn', rel=rel(got, want), off_mag=float(om[:144].abs().max()))
    assert rel(got, want) < 2.3e-3                   # measured 1.1e-3, bar = 2x
    # zero offsets / zero mask logits: the op is 0.5 x the plain 3x3 convolution
    W0 = dict(W)
    W0['d.conv_offset.weight'] = torch.zeros(216, M, 3, 3)
    W0['d.conv_offset.bias'] = torch.zeros(216)
    om0 = torch.zeros(216, h, w, device=dev)
    got0 = planar(ops.conv(cd, ops.dcn_sample(nhwc(x[0], dev), om0, 8)))
    want0 = 0.5 * F.conv2d(x.half().float(), W['d.weight'], None, padding=1)[0] + W['d.bias'].view(-1, 1, 1)
    assert rel(got0, want0) < 1e-3


def test_edvr_pool_upsample_tsa(dev):
    from oracle import refvsr_ir_oracle as iro
    from refvsr_amd import ops
    g = torch.Generator().manual_seed(43)
    x = torch.randn(1, 64, 22, 30, generator=g)
    pp = planar(ops.pool3s2_pair(nhwc(x[0], dev)))
    xh = x.half().float()
    assert maxdiff(pp[:64], iro.pool3s2(xh, 'max')[0]) == 0.0                    # exact on the fp16 values
    assert maxdiff(pp[64:], iro.pool3s2(xh, 'avg')[0]) < 2e-3
    up = planar(ops.up2_bilinear_nhwc16(nhwc(x[0], dev), 2.0))
    assert maxdiff(up, iro.up2_bilinear(xh)[0] * 2) < 4e-3
    al = torch.randn(1, 5, 64, 12, 16, generator=g)
    em = torch.randn(1, 5, 64, 12, 16, generator=g) * 0.3
    er = torch.randn(1, 64, 12, 16, generator=g) * 0.3
    got = planar(ops.tsa_weight([nhwc(al[0, i], dev) for i in range(5)], [nhwc(em[0, i], dev) for i in range(5)], nhwc(er[0], dev)))
    corr = torch.sigmoid((em.half().float() * er.half().float()[:, None]).sum(2))
    want = (al.half().float() * corr[:, :, None]).reshape(1, 320, 12, 16)[0]
    assert maxdiff(got, want) < 4e-3
    f, a, d = [torch.randn(64, 12, 16, generator=g) for _ in range(3)]
    got = planar(ops.tsa_blend(nhwc(f, dev), nhwc(a, dev), nhwc(d, dev)))
    want = f.half().float() * torch.sigmoid(a.half().float()) * 2 + d.half().float()
    assert maxdiff(got, want) < 6e-3


def test_conv_channel_padding_is_zeroed(dev):
    """C = 36 maps (RefVSR_IR) have channel stride 40: the conv epilogues must write zeros into the padding (also through
    the pixel-shuffle store), consumers rely on it."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(47)
    x = torch.randn(36, 19, 27, generator=g)
    xin = nhwc(x, dev)
    assert xin.shape[2] == 40 and float(xin[:, :, 36:].abs().max()) == 0.0
    wt, b = torch.randn(36, 36, 3, 3, generator=g) / 18.0, torch.randn(36, generator=g) * 0.1
    cw = ops.ConvWeights(pack_conv(wt, b, [36]), dev)
    torch.full((64, 64, 64), float('nan'), device=dev).half()              # poison the allocator's free blocks
    torch.cuda.empty_cache()
    for kw in (dict(act=0.2), dict(act=0.2, res=xin), dict(act=1.0, mul=xin, res=xin)):
        y = ops.conv(cw, xin, **kw)
        assert y.shape == (19, 27, 40) and float(y[:, :, 36:].float().abs().max()) == 0.0 and bool(torch.isfinite(y.float()).all())
    want = F.leaky_relu(F.conv2d(x.half().float()[None], wt, b, padding=1), 0.2)[0]
    assert rel(planar(ops.conv(cw, xin, act=0.2))[:36], want) < 1e-3
    ws, bs = torch.randn(144, 36, 3, 3, generator=g) / 18.0, torch.randn(144, generator=g) * 0.1
    cs = ops.ConvWeights(pack_conv(ws, bs, [36], shuffle=True), dev)
    y = ops.conv(cs, xin)
    assert y.shape == (38, 54, 40) and float(y[:, :, 36:].float().abs().max()) == 0.0
    want = F.pixel_shuffle(F.conv2d(x.half().float()[None], ws, bs, padding=1), 2)[0]
    assert rel(planar(y)[:36], want) < 1e-3


# ------------------------------------------------------------------------------------------------ round 4: fused launches
def _rand_conv_weights(cout, cins, seed, dev):
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, sum(cins), 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g) * 0.1
    return ops.ConvWeights(pack_conv(w, b, cins), dev)


@pytest.mark.parametrize('cout', [24, 48])
@pytest.mark.parametrize('h,w', [(8, 32), (19, 45), (40, 70), (135, 240)])
def test_conf_alpha_is_bit_identical_to_the_separate_launches(dev, cout, h, w):
    """refvsr_conf_alpha (cat + [bicubic x2 + clamp] + 2 -> 16 conv + 16 -> C conv [+ max] in one launch, RefVSR.py:47-52,130,
    141-142,147,107-109) against torch.cat + refvsr_resize + refvsr_conv_direct_f32 + refvsr_conv24/48 + refvsr_max2: the same
    fp32 FMA orders and fp16 roundings, so EVERY element must be equal -- tiles at all four frame borders, partial tiles, maps
    smaller than a tile, 24 and 48 output channels."""
    from refvsr_amd import ops
    g = torch.Generator().manual_seed(100 * h + w + cout)
    ca = torch.rand(1, h, w, generator=g).to(dev)
    cb = (torch.rand(1, h, w, generator=g) * 1.2 - 0.1).to(dev)          # values outside [0, 1]: the clamp of the x2 path matters
    w0 = (torch.randn(16, 2, 3, 3, generator=g) * 0.4).to(dev)
    b0 = (torch.randn(16, generator=g) * 0.1).to(dev)
    cw = _rand_conv_weights(cout, [16], 7 + cout, dev)
    assert ops.conf_alpha_ok(cw)
    pair = torch.cat([ca, cb], 0)
    # up = 1 (+ the max by-product)
    a16 = ops.conv_direct(pair, w0, b0, act=0.2, nhwc16_out=True)
    want = ops.conv(cw, a16, act=0.2)
    got, gmax = ops.conf_alpha(ca, cb, 1, w0, b0, cw, want_max=True)
    assert got.shape == want.shape and torch.equal(got, want), 'up=1: %d elements differ' % int((got != want).sum())
    assert torch.equal(gmax, ops.max2(ca, cb))
    # up = 2
    pair_up = ops.bicubic_scale(pair, 2, clamp01=True)
    a16 = ops.conv_direct(pair_up, w0, b0, act=0.2, nhwc16_out=True)
    want = ops.conv(cw, a16, act=0.2)
    got = ops.conf_alpha(ca, cb, 2, w0, b0, cw)
    assert got.shape == want.shape and torch.equal(got, want), 'up=2: %d elements differ' % int((got != want).sum())
    report('conf_alpha %dx%d C=%d' % (h, w, cout), equal=1)


@pytest.mark.parametrize('h,w,hin,win', [(17, 29, 34, 58), (64, 96, 128, 192), (33, 40, 33, 40)])
def test_warp_up2_is_bit_identical(dev, h, w, hin, win):
    """refvsr_warp_nhwc16_up2 == refvsr_resize(BILINEAR_AC x2, * 2) + refvsr_warp_nhwc16 (RefVSR.py:220,254,259), bit for bit;
    the last case is the :254 quirk (an LR-size map sampled on the 2x grid)."""
    from refvsr_amd import ops
    g = torch.Generator().manual_seed(h * 1000 + w)
    x = nhwc(torch.randn(24, hin, win, generator=g), dev)
    fl = (torch.randn(2, h, w, generator=g) * 3).to(dev)
    want = ops.warp_nhwc16(x, ops.flow_up2(fl))
    got = ops.warp_nhwc16_up2(x, fl)
    assert got.shape == (2 * h, 2 * w, 24) and torch.equal(got, want)


def test_batched_conv_and_spynet_level_input_equal_the_single_launches(dev):
    """RefvsrConv.batch (blockIdx.y = image) and refvsr_spynet_level_input_batch: image b of the batched launch equals the
    single-image launch bit for bit -- streamed 7x7 (plain fp16 and hi + lo, mt = 1 and 2), resident 7x7 8 -> 32, the planar
    flow head with its planar residual."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(5)
    for (co, ci, hh, ww, kw) in [(32, 8, 18, 30, {}), (64, 32, 18, 30, dict(hi_only=True)), (64, 32, 9, 15, dict(mt=1, hi_only=True)),
                                 (32, 64, 36, 60, {}), (16, 32, 40, 72, dict(hi_only=True))]:
        w_ = torch.randn(co, ci, 7, 7, generator=g) * 0.05
        b_ = torch.randn(co, generator=g) * 0.1
        cw = ops.ConvWeights(pack_conv(w_, b_, [ci], **kw), dev)
        xs = [nhwc(torch.randn(ci, hh, ww, generator=g), dev) for _ in range(3)]
        got = ops.conv(cw, torch.stack(xs, 0).contiguous(), act=0.0, batch=3)
        for b in range(3):
            assert torch.equal(got[b], ops.conv(cw, xs[b], act=0.0)), (co, ci, b)
    w_ = torch.randn(2, 16, 7, 7, generator=g) * 0.05
    cw = ops.ConvWeights(pack_conv(w_, torch.randn(2, generator=g), [16]), dev)
    xs = [nhwc(torch.randn(16, 20, 36, generator=g), dev) for _ in range(2)]
    rp = [torch.randn(2, 20, 36, generator=g).to(dev) for _ in range(2)]
    got = ops.conv(cw, torch.stack(xs, 0).contiguous(), planar_out=True, res_planar=torch.stack(rp, 0).contiguous(), batch=2)
    for b in range(2):
        assert torch.equal(got[b], ops.conv(cw, xs[b], planar_out=True, res_planar=rp[b]))
    for nb in (2, 8):                                      # (eight pairs per launch since ABI 11: the flows of a frame group)
        refs = [torch.rand(3, 20, 36, generator=g).to(dev) for _ in range(nb)]
        sups = [torch.rand(3, 20, 36, generator=g).to(dev) for _ in range(nb)]
        fp = (torch.randn(nb, 2, 10, 18, generator=g) * 2).to(dev)
        for flow_prev in (None, fp):
            x8, fup = ops.spynet_level_input_batch(refs, sups, flow_prev)
            for b in range(nb):
                x1, f1 = ops.spynet_level_input(refs[b], sups[b], None if flow_prev is None else flow_prev[b].contiguous())
                assert torch.equal(x8[b], x1) and torch.equal(fup[b], f1)


@pytest.mark.parametrize('h,w', [(19, 45), (64, 96), (135, 240)])
def test_resblock24_store_modes_are_bit_identical(dev, h, w):
    """refvsr_set_resblock24_store: the 16-byte stores after the v_permlane16_swap exchange (1, the default) write exactly the
    bytes of the 8-byte stores (0) -- border tiles, partial tiles, ReLU and leaky blocks, 8- and 16-wave shapes."""
    from refvsr_amd import hip, ops
    g = torch.Generator().manual_seed(h + w)
    raw = []
    for _ in range(3):
        ws = [torch.randn(24, 24, 3, 3, generator=g) / (24 * 9) ** 0.5 for _ in range(2)]
        bs = [torch.randn(24, generator=g) * 0.1 for _ in range(2)]
        raw.append(((ws[0], bs[0]), (ws[1], bs[1])))
    ch = ops.Resblock24Chain(raw, dev)
    x = nhwc(torch.randn(24, h, w, generator=g), dev)
    lib = hip.lib()
    try:
        for waves in (8, 16):
            lib.refvsr_set_resblock24_waves(waves)
            for act in (0.0, 0.2):
                lib.refvsr_set_resblock24_store(0)
                want = ops.resblock24_chain(ch, x, act)
                for mode in (1,):
                    lib.refvsr_set_resblock24_store(mode)
                    got = ops.resblock24_chain(ch, x, act)
                    assert torch.equal(got, want), 'store mode %d, %d waves, act %.1f: %d elements differ' % (mode, waves, act, int((got != want).sum()))
    finally:
        lib.refvsr_set_resblock24_waves(0)
        assert lib.refvsr_set_resblock24_store(2) != 0                      # (round 4's write-through mode is gone) rejected, mode unchanged
        lib.refvsr_set_resblock24_store(int(os.environ.get('REFVSR_RB24_STORE', str(RB24_STORE_DEFAULT))))


@pytest.mark.parametrize('h,w', [(8, 32), (19, 45), (64, 96), (135, 240)])
def test_resblock48_chain_equals_two_conv48_launches(dev, h, w):
    """refvsr_resblock48_chain (one launch per block, the two 84 KB weight sets swapped per tile by LDS-DMA, intermediate tile
    in LDS) against the round-3 path -- refvsr_conv48 with the activation, refvsr_conv48 with the residual -- on the same
    packed weights: same K order, same fp16 rounding of the intermediate, residual added after the accumulation => every
    element equal.  Border tiles on all sides, partial tiles, a map of one tile, persistent workgroups with two tiles
    (135 x 240 has 136 tiles; 270 x 480 runs in the end-to-end tests), ReLU and leaky blocks, chains of three."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(48 * h + w)
    pairs = []
    for _ in range(3):
        cws = []
        for _ in range(2):
            w_ = torch.randn(48, 48, 3, 3, generator=g) / (48 * 9) ** 0.5
            b_ = torch.randn(48, generator=g) * 0.1
            cws.append(ops.ConvWeights(pack_conv(w_, b_, [48]), dev))
            assert cws[-1].blob24 is not None and cws[-1].raw is not None
        pairs.append(tuple(cws))
    ch = ops.Resblock48Chain(pairs, dev)
    x = nhwc(torch.randn(48, h, w, generator=g), dev)
    for act in (0.0, 0.2):
        want = x
        for c1, c2 in pairs:
            t = ops.conv(c1, want, act=act)
            want = ops.conv(c2, t, res=want)
        got = ops.resblock48_chain(ch, x, act)
        assert got.shape == want.shape
        assert torch.equal(got, want), 'act %.1f: %d of %d elements differ, max %.3e' % (
            act, int((got != want).sum()), got.numel(), float((got.float() - want.float()).abs().max()))
    report('resblock48 %dx%d' % (h, w), equal=1)


@pytest.mark.parametrize('c', [24, 48])
@pytest.mark.parametrize('bh,bw,scale', [(19, 45, 4), (64, 96, 4), (33, 50, 2), (2, 8, 4)])
def test_conv_last_fused_head(dev, c, bh, bw, scale):
    """refvsr_conv_last (conv_last 3x3 C -> 3 + the bicubic base + both clamps in one launch, RefVSR.py:92,118,288,297) against
    refvsr_resize (bicubic, clamped) + refvsr_conv_mfma's planar mode: the base values are the same FMA chains (shared device
    function), the conv is summed in conv24's K order instead of the generic one => equal to fp32 rounding (bar 2e-5 on values in
    [0, 1]); against an fp32 torch restatement of the whole head to 2e-3 (fp16 activations, hi + lo weights).  Border and partial
    tiles, x4 and x2, 24 and 48 channels, values that hit both clamps."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv, pack_conv_last
    g = torch.Generator().manual_seed(c + bh * bw + scale)
    h, w = bh * scale, bw * scale
    w_ = torch.randn(3, c, 3, 3, generator=g) * 0.03          # conv std ~0.45-0.6 around the base: both clamps hit, most values inside
    b_ = torch.randn(3, generator=g) * 0.1
    xf = torch.randn(c, h, w, generator=g)
    base = (torch.rand(3, bh, bw, generator=g) * 1.2 - 0.1).to(dev)
    x = nhwc(xf, dev)
    cw = ops.ConvWeights(pack_conv(w_, b_, [c]), dev)
    want = ops.conv(cw, x, planar_out=True, res_planar=ops.bicubic_scale(base, scale, clamp01=True), clamp=(0.0, 1.0))
    assert ops.conv_last_ok(c, h, w)
    got = ops.conv_last(pack_conv_last(w_, b_).to(dev), x, base)
    assert got.shape == want.shape == (3, h, w)
    e = maxdiff(got, want)
    ref = (F.conv2d(planar(x, c)[None], w_, b_, padding=1)[0] +
           F.interpolate(base.cpu()[None], scale_factor=scale, mode='bicubic', align_corners=False)[0].clamp(0, 1)).clamp(0, 1)
    e_ref = maxdiff(got.cpu(), ref)
    report('conv_last %dx%d x%d C=%d' % (bh, bw, scale, c), vs_generic=e, vs_torch=e_ref)
    assert e < 2e-5 and e_ref < 2e-3
    assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0 and float((got == 0).float().mean()) > 0.01 and float((got == 1).float().mean()) > 0.01


@pytest.mark.parametrize('h,w,act,post,use_mul,use_res', [(8, 32, 0.2, 1.0, False, False), (19, 45, 0.2, 1.0, False, False),
                                                          (33, 70, 1.0, 1.0, False, False), (61, 130, 0.2, 1.0, True, True),
                                                          (7, 5, 1.0, 0.2, False, True), (270, 480, 0.2, 1.0, False, False),
                                                          (540, 960, 0.2, 1.0, False, False)])
def test_conv48_two_source_channel_halves(dev, h, w, act, post, use_mul, use_res):
    """refvsr_conv48 with two 48-channel sources (feat_fusion*.0 / feat_fusion2_1 / fusion_UP of the mid_channels = 48 models,
    RefVSR.py:53-62,87): the output channels are computed in two halves of 24 on blockIdx.y (NCG = 12 K plan, 81 KB of weights per
    half resident) and stored into the 48-channel map.  Against torch fp32 on the same fp16 maps and against the runtime-generic
    streamed kernel it replaces (same arithmetic up to fp32 summation order); every epilogue combination, border / partial tiles,
    maps smaller than a tile, sixteen waves with ONE pixel group each (odd group count per wave)."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(h * 5 + w)
    wt = torch.randn(48, 96, 3, 3, generator=g) / (96 * 9) ** 0.5
    b = torch.randn(48, generator=g) * 0.1
    x = torch.randn(1, 96, h, w, generator=g)
    cw = ops.ConvWeights(pack_conv(wt, b, [48, 48]), dev)
    assert cw.blob24 is not None and cw.blob24.numel() == 2 * (27 * 3 * 1024 + 128)
    s0, s1 = nhwc(x[0, :48], dev), nhwc(x[0, 48:], dev)
    mul = torch.rand(48, h, w, generator=g) if use_mul else None
    res = torch.randn(48, h, w, generator=g) if use_res else None
    kw = dict(act=act, post=post, mul=nhwc(mul, dev) if use_mul else None, res=nhwc(res, dev) if use_res else None)
    got = ops.conv(cw, s0, s1, **kw)
    blob, cw.blob24 = cw.blob24, None                      # the same call through the generic (streamed) kernel
    gen = ops.conv(cw, s0, s1, **kw)
    cw.blob24 = blob
    want = F.leaky_relu(F.conv2d(x.half().float(), wt, b, padding=1), act)[0]
    if use_mul:
        want = want * mul.half().float()
    if use_res:
        want = want + res.half().float()
    want = F.leaky_relu(want, post)
    e, d = rel(planar(got), want), maxdiff(planar(got), planar(gen))
    report('conv48 48+48 %dx%d act%.1f%s%s' % (h, w, act, ' mul' if use_mul else '', ' res' if use_res else ''), rel=e, vs_generic=d)
    assert got.shape == gen.shape == (h, w, 48)
    assert e < 1e-3
    assert d < 4e-3                                        # an fp16 ulp where the fp32 sums round differently


@pytest.mark.parametrize('h,w', [(8, 32), (19, 45), (64, 96), (270, 480)])
def test_conv48_input_conv_8_plus_48(dev, h, w):
    """refvsr_conv48 on cat([lr (3 channels in an 8-channel map), feat (48)]) -- the input conv of ResidualBlocksWithInputConv for
    mid_channels = 48 (RefVSR.py:340-343) -- on the NCG = 7 K plan (two K-steps per tap, one zero block each), 108 KB of weights
    resident, sixteen waves with one pixel group each; against torch fp32 and the generic kernel it replaces."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(h + w)
    wt = torch.randn(48, 51, 3, 3, generator=g) / (51 * 9) ** 0.5
    b = torch.randn(48, generator=g) * 0.1
    x = torch.randn(1, 51, h, w, generator=g)
    cw = ops.ConvWeights(pack_conv(wt, b, [3, 48]), dev)
    assert cw.blob24 is not None and cw.blob24.numel() == 18 * 6 * 1024 + 256
    s0, s1 = nhwc(x[0, :3], dev, 8), nhwc(x[0, 3:], dev)
    got = ops.conv(cw, s0, s1, act=0.1)
    blob, cw.blob24 = cw.blob24, None
    gen = ops.conv(cw, s0, s1, act=0.1)
    cw.blob24 = blob
    want = F.leaky_relu(F.conv2d(x.half().float(), wt, b, padding=1), 0.1)[0]
    e, d = rel(planar(got), want), maxdiff(planar(got), planar(gen))
    report('conv48 8+48 %dx%d' % (h, w), rel=e, vs_generic=d)
    assert got.shape == gen.shape == (h, w, 48) and e < 1e-3 and d < 4e-3


@pytest.mark.parametrize('bh,bw,scale', [(19, 45, 4), (2, 8, 4), (33, 50, 2), (64, 96, 4), (270, 480, 4)])
def test_conv_hr_last_fused_tail(dev, bh, bw, scale):
    """refvsr_conv_hr_last (conv_hr + LeakyReLU + conv_last + bicubic base + clamps in one launch on the fused block's skeleton,
    RefVSR.py:91-92,116-118,288,297) against refvsr_conv24 + refvsr_conv_last: the intermediate tile holds the same fp16 values
    the HR map would (same K plan, same rounding), the head is summed in another K order => equal to fp32 rounding (bar 2e-5);
    against fp32 torch to 2e-3.  8-wave 8 x 32 tiles and (1080 x 1920) the 16-wave 16 x 32 tiles, border / partial tiles."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv, pack_conv_hr_last, pack_conv_last
    g = torch.Generator().manual_seed(7 * bh + bw + scale)
    h, w = bh * scale, bw * scale
    w1 = torch.randn(24, 24, 3, 3, generator=g) / (24 * 9) ** 0.5
    b1 = torch.randn(24, generator=g) * 0.1
    w2 = torch.randn(3, 24, 3, 3, generator=g) * 0.04
    b2 = torch.randn(3, generator=g) * 0.1
    xf = torch.randn(24, h, w, generator=g)
    base = (torch.rand(3, bh, bw, generator=g) * 1.2 - 0.1).to(dev)
    x = nhwc(xf, dev)
    t = ops.conv(ops.ConvWeights(pack_conv(w1, b1, [24]), dev), x, act=0.1)
    want = ops.conv_last(pack_conv_last(w2, b2).to(dev), t, base)
    got = ops.conv_hr_last(pack_conv_hr_last(w1, b1, w2, b2).to(dev), x, base, act=0.1)
    assert got.shape == want.shape == (3, h, w)
    e = maxdiff(got, want)
    e_ref = float('nan')
    if h * w <= 512 * 512:
        tt = F.leaky_relu(F.conv2d(planar(x, 24)[None], w1, b1, padding=1), 0.1).half().float()
        ref = (F.conv2d(tt, w2, b2, padding=1)[0] + F.interpolate(base.cpu()[None], scale_factor=scale, mode='bicubic', align_corners=False)[0].clamp(0, 1)).clamp(0, 1)
        e_ref = maxdiff(got.cpu(), ref)
        assert e_ref < 2e-3
    report('conv_hr_last %dx%d x%d' % (bh, bw, scale), vs_two_launches=e, vs_torch=e_ref)
    assert e < 2e-5
    assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize('cin,size', [(64, (270, 480)), (64, (19, 45)), (128, (68, 120)), (64, (1, 3))])
def test_conv1x1_f32_map(dev, cin, size):
    """refvsr_conv1x1_f32 (the map64 / map128 block of the matching's feature extractor, RefVSR_/attention.py:41-42) against
    F.conv2d + leaky_relu in float64 and against the generic conv's fp32 mode it replaces: fp32 accuracy (the arg-max of the
    matching is decided on these features)."""
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    g = torch.Generator().manual_seed(7)
    h, w = size
    x = (torch.randn(cin, h, w, generator=g).abs() * 2.0).to(dev)
    wt = (torch.randn(16, cin, 1, 1, generator=g) * 0.2).to(dev)
    b = (torch.randn(16, generator=g) * 0.1).to(dev)
    xh = ops.pack_nhwc32(x, cin)
    got = ops.conv1x1_f32(xh, wt.reshape(16, cin).contiguous(), b, 0.2)
    want = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x.double()[None], wt.double(), b.double()), 0.2)[0]
    old = ops.conv(ops.ConvWeights(pack_conv(wt.cpu(), b.cpu(), [cin], f32=True), dev), xh, act=0.2, planar_out=True)
    scale = float(want.abs().max())
    e_new, e_old = maxdiff(got.double(), want) / scale, maxdiff(old.double(), want) / scale
    report('conv1x1_f32 %d->16 %dx%d' % (cin, h, w), rel_vs_f64=e_new, generic_rel_vs_f64=e_old)
    assert got.shape == (16, h, w) and e_new < 2e-6 and e_old < 2e-6


# ---- multi-map launches (ABI 11) ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B', [2, 3, 4])
@pytest.mark.parametrize('h,w', [(8, 32), (19, 45), (64, 96), (270, 480)])
def test_multimap_launches_equal_the_single_map_launches(dev, B, h, w):
    """refvsr_*_batch: map b of a multi-map launch == the single-map launch on map b, bit for bit -- the fused 24-channel block
    chain (n = 1, 2, 3: all scratch configurations; ReLU and leaky), every conv24 input shape with every epilogue operand (the
    operands are separately allocated maps, as the per-frame cached maps of the engine are), the pixel-shuffle conv, both
    confidence fusions with the max by-product, the three warps.  Sizes: smaller than a tile, partial tiles, one tile per
    workgroup x B (270 x 480: the case the launches exist for)."""
    from refvsr_amd import ops
    g = torch.Generator().manual_seed(1000 * B + 10 * h + w)
    rn = lambda c, hh=h, ww=w: nhwc(torch.randn(c, hh, ww, generator=g), dev)
    # fused block chain
    for n, act in ((1, 0.0), (2, 0.2), (3, 0.0)):
        raw = [((torch.randn(24, 24, 3, 3, generator=g) * 0.07, torch.randn(24, generator=g) * 0.1),
                (torch.randn(24, 24, 3, 3, generator=g) * 0.07, torch.randn(24, generator=g) * 0.1)) for _ in range(n)]
        ch = ops.Resblock24Chain(raw, dev)
        xs = [rn(24) for _ in range(B)]
        got = ops.resblock24_chain_b(ch, xs, act)
        assert got.shape == (B, h, w, 24)
        for b in range(B):
            assert torch.equal(got[b], ops.resblock24_chain(ch, xs[b], act)), ('rb24', n, b)
    # the 48-channel fused block chain (ABI 12)
    for n, act in ((1, 0.0), (2, 0.2), (3, 0.0)):
        raw = [((torch.randn(48, 48, 3, 3, generator=g) * 0.05, torch.randn(48, generator=g) * 0.1),
                (torch.randn(48, 48, 3, 3, generator=g) * 0.05, torch.randn(48, generator=g) * 0.1)) for _ in range(n)]
        ch = ops.Resblock48Chain(raw, dev)
        xs = [rn(48) for _ in range(B)]
        got = ops.resblock48_chain_b(ch, xs, act)
        assert got.shape == (B, h, w, 48)
        for b in range(B):
            assert torch.equal(got[b], ops.resblock48_chain(ch, xs[b], act)), ('rb48', n, b)
    # conv24: every input shape x epilogue operands
    for cins, act, post, use_mul, use_res in (([24], 0.2, 1.0, True, True), ([16], 0.2, 1.0, False, False), ([8, 24], 0.1, 1.0, False, False),
                                              ([24, 24], 0.2, 0.2, False, True), ([24], 1.0, 1.0, False, True)):
        cw = _rand_conv_weights(24, cins, 3 + len(cins) + cins[0], dev)
        assert cw.blob24 is not None
        s0 = [rn(cins[0]) for _ in range(B)]
        s1 = [rn(cins[1]) for _ in range(B)] if len(cins) > 1 else None
        muls = [nhwc(torch.rand(24, h, w, generator=g), dev) for _ in range(B)] if use_mul else None
        ress = [rn(24) for _ in range(B)] if use_res else None
        got = ops.conv_b(cw, s0, s1, act=act, muls=muls, ress=ress, post=post)
        assert got.shape == (B, h, w, 24)
        for b in range(B):
            want = ops.conv(cw, s0[b], None if s1 is None else s1[b], act=act, mul=None if muls is None else muls[b],
                            res=None if ress is None else ress[b], post=post)
            assert torch.equal(got[b], want), ('conv24', cins, b)
    # pixel-shuffle conv (upsample1)
    gq = torch.Generator().manual_seed(77)
    from refvsr_amd.packing import pack_conv
    cws = ops.ConvWeights(pack_conv(torch.randn(96, 24, 3, 3, generator=gq) * 0.07, torch.randn(96, generator=gq) * 0.1, [24], True), dev)
    assert cws.blob24 is not None and cws.shuffle
    xs = [rn(24) for _ in range(B)]
    got = ops.conv_b(cws, xs)
    assert got.shape == (B, 2 * h, 2 * w, 24)
    for b in range(B):
        assert torch.equal(got[b], ops.conv(cws, xs[b])), ('shuffle', b)
    # confidence fusions
    w0 = (torch.randn(16, 2, 3, 3, generator=g) * 0.4).to(dev)
    b0 = (torch.randn(16, generator=g) * 0.1).to(dev)
    cwa = _rand_conv_weights(24, [16], 31, dev)
    cas = [torch.rand(1, h, w, generator=g).to(dev) for _ in range(B)]
    cbs = [(torch.rand(1, h, w, generator=g) * 1.2 - 0.1).to(dev) for _ in range(B)]
    got, gmax = ops.conf_alpha_b(cas, cbs, 1, w0, b0, cwa, want_max=True)
    got2 = ops.conf_alpha_b(cas, cbs, 2, w0, b0, cwa)
    for b in range(B):
        a1, m1 = ops.conf_alpha(cas[b], cbs[b], 1, w0, b0, cwa, want_max=True)
        assert torch.equal(got[b], a1) and torch.equal(gmax[b], m1), ('conf1', b)
        assert torch.equal(got2[b], ops.conf_alpha(cas[b], cbs[b], 2, w0, b0, cwa)), ('conf2', b)
    # warps
    fls = [(torch.randn(2, h, w, generator=g) * 3).to(dev) for _ in range(B)]
    xs = [rn(24) for _ in range(B)]
    xu = [rn(24, 2 * h, 2 * w) for _ in range(B)]
    got = ops.warp_nhwc16_b(xs, fls)
    gup = ops.warp_nhwc16_up2_b(xu, fls)
    gq_ = ops.warp_nhwc16_up2_b(xs, fls)                      # the :254 quirk: an LR-size map sampled on the 2x grid
    gpl = ops.warp_planar_b(cas, fls)
    for b in range(B):
        assert torch.equal(got[b], ops.warp_nhwc16(xs[b], fls[b])), ('warp', b)
        assert torch.equal(gup[b], ops.warp_nhwc16_up2(xu[b], fls[b])), ('warp_up2', b)
        assert torch.equal(gq_[b], ops.warp_nhwc16_up2(xs[b], fls[b])), ('warp_up2 quirk', b)
        assert torch.equal(gpl[b], ops.warp_planar(cas[b], fls[b])), ('warp_planar', b)
    report('multimap B=%d %dx%d' % (B, h, w), equal=1)
