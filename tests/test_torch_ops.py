"""`torch.ops.refvsr.*`: the PyTorch-ROCm custom-op registration of the hot path (refvsr_amd/torch_ops.py).
CPU part: the ops exist in the dispatcher with the documented schemas and infer shapes / dtypes on fake tensors (what
torch.compile / make_fx need).  GPU part: results equal the direct C-ABI calls bit for bit, and torch's own `opcheck`
accepts the registrations."""
import pytest
import torch


def test_ops_are_registered_with_schemas():
    import refvsr_amd.torch_ops as t
    for name in t.OP_NAMES:
        assert hasattr(torch.ops.refvsr, name), name
    s = str(torch.ops.refvsr.conv_mfma.default._schema)
    assert s.startswith('refvsr::conv_mfma(Tensor wpack, Tensor bias, SymInt[] meta, Tensor src0, Tensor? src1')
    assert str(torch.ops.refvsr.match_argmax.default._schema) == 'refvsr::match_argmax(Tensor lr_feat, Tensor ref_feat) -> (Tensor, Tensor)'


def test_fake_tensor_shape_inference():
    import refvsr_amd.torch_ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        dev = 'cuda'
        x = torch.empty((20, 30, 24), dtype=torch.float16, device=dev)
        fl = torch.empty((2, 40, 60), device=dev)
        w = torch.empty(8, device=dev)
        assert torch.ops.refvsr.warp(x, fl).shape == (40, 60, 24)                                    # LR input / 2x flow form
        assert torch.ops.refvsr.resblock(w, w, w, w, 7, x, 0.0, 1.0).shape == x.shape
        assert torch.ops.refvsr.resblock24_chain(torch.empty((3, 43264), dtype=torch.uint8, device=dev), x, 0.0).shape == x.shape
        y = torch.ops.refvsr.conv24(w, torch.empty((20, 30, 8), dtype=torch.float16, device=dev), x, None, None, 0.1, 1.0)
        assert y.shape == (20, 30, 24) and y.dtype == torch.float16
        assert torch.ops.refvsr.conv_shuffle2(torch.empty(86528, dtype=torch.uint8, device=dev), x, 0.1).shape == (40, 60, 24)
        y = torch.ops.refvsr.conv_mfma(w, w, [96, 3, 1, 0, 7, 3, 24, 0], x, None, None, None, None, 1, 1.0, 1.0, False, 0.0, 0.0, 0.0)
        assert y.shape == (40, 60, 24) and y.dtype == torch.float16                                   # pixel-shuffle weights
        y = torch.ops.refvsr.conv_mfma(w, w, [3, 3, 0, 0, 7, 1, 24, 0], x, None, None, None, None, 1, 1.0, 1.0, True, 0.0, 0.0, 1.0)
        assert y.shape == (3, 20, 30) and y.dtype == torch.float32
        y = torch.ops.refvsr.conv_mfma(w, w, [32, 5, 0, 0, 7, 2, 32, 32], x, x, None, None, None, 2, 0.2, 1.0, False, 0.0, 0.0, 0.0)
        assert y.shape == (10, 15, 32)
        # C = 36 maps (RefVSR_IR) carry a channel stride of 40: the fake impl rounds like ops.conv
        x40 = torch.empty((20, 30, 40), dtype=torch.float16, device=dev)
        y = torch.ops.refvsr.conv_mfma(w, w, [36, 3, 0, 0, 12, 3, 40, 0], x40, None, None, None, None, 1, 1.0, 1.0, False, 0.0, 0.0, 0.0)
        assert y.shape == (20, 30, 40)
        y = torch.ops.refvsr.conv_mfma(w, w, [144, 3, 1, 0, 12, 3, 40, 0], x40, None, None, None, None, 1, 1.0, 1.0, False, 0.0, 0.0, 0.0)
        assert y.shape == (40, 60, 40)                                                                # shuffle: 144 / 4 = 36 -> 40
        conf, idx = torch.ops.refvsr.match_argmax(torch.empty((16, 20, 30), device=dev), torch.empty((16, 10, 15), device=dev))
        assert conf.shape == (1, 20, 30) and idx.shape == (600,) and idx.dtype == torch.int32
        assert torch.ops.refvsr.block_gather(x, idx, 20, 30, 2).shape == (40, 60, 24)
        assert torch.ops.refvsr.aligned_sample(x, torch.empty((3, 10, 15), device=dev), 2).shape == x.shape
        assert torch.ops.refvsr.resize(torch.empty((3, 20, 30), device=dev), 80, 120, 0, 0.25, 0.25, True).shape == (3, 80, 120)
        x8, fu = torch.ops.refvsr.spynet_level_input(torch.empty((3, 32, 32), device=dev), torch.empty((3, 32, 32), device=dev), None)
        assert x8.shape == (32, 32, 8) and x8.dtype == torch.float16 and fu.shape == (2, 32, 32)
        y = torch.ops.refvsr.resblock24_chain_batch(torch.empty((3, 43264), dtype=torch.uint8, device=dev), [x, x, x], 0.0)
        assert y.shape == (3, 20, 30, 24) and y.dtype == torch.float16                                # multi-map launches: [B, ...] out
        y = torch.ops.refvsr.conv24_batch(w, [x, x], [x, x], [], [x, x], 0.2, 1.0)
        assert y.shape == (2, 20, 30, 24) and y.dtype == torch.float16
        assert torch.ops.refvsr.warp_batch([x, x], [fl, fl]).shape == (2, 40, 60, 24)
        assert torch.ops.refvsr.pack_nhwc16(torch.empty((3, 20, 30), device=dev), 8).shape == (20, 30, 8)
        assert torch.ops.refvsr.unpack_nhwc16(x, 24).shape == (24, 20, 30)


@pytest.mark.gpu
def test_torch_library_ops_match_direct_calls():
    import refvsr_amd.torch_ops as t
    from refvsr_amd import ops
    from refvsr_amd.packing import pack_conv
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    C, h, w = 24, 36, 52
    x = ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev))
    mk = lambda co, cins, **kw: ops.ConvWeights(pack_conv(torch.randn(co, sum(cins), 3, 3, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1, cins, **kw), dev)
    c1, c2, cs, cp = mk(C, [C]), mk(C, [C]), mk(4 * C, [C], shuffle=True), mk(3, [C])
    R = torch.ops.refvsr
    # the specialised kernels are their own ops; conv_mfma is the generic kernel (direct call with the conv24 blob set aside)
    assert c1.blob24 is not None
    assert torch.equal(R.conv24(c1.blob24, x, None, None, x, 0.2, 1.0), ops.conv(c1, x, act=0.2, res=x))
    ch = ops.Resblock24Chain([(c1, c2), (c2, c1)], dev)
    assert torch.equal(R.resblock24_chain(ch.blobs, x, 0.0), ops.resblock24_chain(ch, x, 0.0))
    blob, c1.blob24 = c1.blob24, None
    assert torch.equal(R.conv_mfma(c1.wpack, c1.bias, t.conv_meta(c1), x, None, None, x, None, 1, 0.2, 1.0, False, 0.0, 0.0, 0.0),
                       ops.conv(c1, x, act=0.2, res=x))
    c1.blob24 = blob
    assert cs.blob24 is not None and torch.equal(R.conv_shuffle2(cs.blob24, x, 0.1), ops.conv(cs, x, act=0.1))      # the specialised pixel-shuffle conv
    blob, cs.blob24 = cs.blob24, None
    assert torch.equal(R.conv_mfma(cs.wpack, cs.bias, t.conv_meta(cs), x, None, None, None, None, 1, 1.0, 1.0, False, 0.0, 0.0, 0.0), ops.conv(cs, x))
    cs.blob24 = blob
    base = torch.rand(3, h, w, generator=g).to(dev)
    assert torch.equal(R.conv_mfma(cp.wpack, cp.bias, t.conv_meta(cp), x, None, None, None, base, 1, 1.0, 1.0, True, 0.0, 0.0, 1.0),
                       ops.conv(cp, x, planar_out=True, res_planar=base, clamp=(0.0, 1.0)))
    assert torch.equal(R.resblock(c1.wpack, c1.bias, c2.wpack, c2.bias, c1.ksteps, x, 0.0, 1.0), ops.resblock(c1, c2, x, act=0.0))
    fl = (torch.randn(2, h, w, generator=g) * 2).to(dev)
    assert torch.equal(R.warp(x, fl), ops.warp_nhwc16(x, fl))
    # multi-map launches through the dispatcher: map b == the single-map op on map b
    x2 = ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev))
    fl2 = (torch.randn(2, h, w, generator=g) * 2).to(dev)
    yb = R.resblock24_chain_batch(ch.blobs, [x, x2], 0.0)
    assert yb.shape == (2, h, w, C) and torch.equal(yb[0], ops.resblock24_chain(ch, x, 0.0)) and torch.equal(yb[1], ops.resblock24_chain(ch, x2, 0.0))
    yb = R.conv24_batch(c1.blob24, [x, x2], [], [], [x2, x], 0.2, 1.0)
    assert torch.equal(yb[0], ops.conv(c1, x, act=0.2, res=x2)) and torch.equal(yb[1], ops.conv(c1, x2, act=0.2, res=x))
    yb = R.warp_batch([x, x2], [fl, fl2])
    assert torch.equal(yb[0], ops.warp_nhwc16(x, fl)) and torch.equal(yb[1], ops.warp_nhwc16(x2, fl2))
    torch.library.opcheck(R.resblock24_chain_batch, (ch.blobs, [x, x2], 0.0), test_utils=('test_schema', 'test_faketensor'))
    torch.library.opcheck(R.conv24_batch, (c1.blob24, [x, x2], [], [], [x2, x], 0.2, 1.0), test_utils=('test_schema', 'test_faketensor'))
    torch.library.opcheck(R.warp_batch, ([x, x2], [fl, fl2]), test_utils=('test_schema', 'test_faketensor'))
    lr_f, ref_f = torch.randn(16, h, w, generator=g).to(dev), torch.randn(16, h // 2, w // 2, generator=g).to(dev)
    conf, idx = R.match_argmax(lr_f, ref_f)
    lr_rows, inv_lr, l_lo = ops.match_patches(lr_f, 512, want_lo=True)
    ref_rows, inv_ref, r_lo = ops.match_patches(ref_f, 256, want_lo=True)
    cand, cv = ops.match_top2(ref_rows, (h // 2) * (w // 2), lr_rows, h * w, 1)
    c0, i0, _ = ops.match_refine(lr_f, ref_f, inv_lr, inv_ref, cand, cv, ops.MATCH_EXACT_MARGIN, (lr_rows, l_lo), (ref_rows, r_lo))
    assert torch.equal(conf.view(-1), c0) and torch.equal(idx, i0)
    assert torch.equal(R.block_gather(x, idx, h, w, 2), ops.block_gather_nhwc16(x, idx, h, w, 2))
    with pytest.raises(RuntimeError, match='ksteps mismatch'):          # the library's own argument check surfaces as RuntimeError
        R.resblock(c1.wpack, c1.bias, c2.wpack, c2.bias, c1.ksteps + 1, x, 0.0, 1.0)
    # torch's own checker of custom-op registrations (schema, fake impl vs real output metadata, functionalisation)
    torch.library.opcheck(R.warp, (x, fl), test_utils=('test_schema', 'test_faketensor'))
    torch.library.opcheck(R.resblock, (c1.wpack, c1.bias, c2.wpack, c2.bias, c1.ksteps, x, 0.0, 1.0), test_utils=('test_schema', 'test_faketensor'))
    torch.library.opcheck(R.match_argmax, (lr_f, ref_f), test_utils=('test_schema', 'test_faketensor'))
    torch.library.opcheck(R.conv24, (c1.blob24, x, None, None, x, 0.2, 1.0), test_utils=('test_schema', 'test_faketensor'))
    torch.library.opcheck(R.resblock24_chain, (ch.blobs, x, 0.0), test_utils=('test_schema', 'test_faketensor'))
    torch.library.opcheck(R.conv_shuffle2, (cs.blob24, x, 0.1), test_utils=('test_schema', 'test_faketensor'))
