"""Host logic on CPU: config mirror, state-dict contract, weight packing, model shell, clip windows,
shard partitioning."""
import collections

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from refvsr_amd import CONFIG_NAMES, get_config, make_state_dict, state_spec
from refvsr_amd import shard, weights
from refvsr_amd.packing import choose_mt, pack_conv
from refvsr_amd.synth import make_clip, window_indices


def test_configs_match_reference_fields():
    g = load_golden('state_spec')      # produced from the reference's own config modules
    for name in CONFIG_NAMES:
        cfg = get_config('p', 'm', name)
        assert cfg.frame_num == int(g[name + '/frame_num'])
        spec = state_spec(cfg)
        assert len(spec) == int(g[name + '/ntensors'])
        assert weights.num_params(cfg) == int(g[name + '/nparams'])
        assert weights.spec_checksum(cfg) == int(g[name + '/spec_crc'])
    c = get_config('p', 'm', 'config_RefVSR_small_L1')
    assert (c.mid_channels, c.num_blocks, c.matching_ksize, c.reset_branch, c.scale) == (24, 24, 2, 26, 4)
    c = get_config('p', 'm', 'config_RefVSR_MFID_8K')
    assert (c.mid_channels, c.num_blocks, c.matching_ksize, c.reset_branch, c.flag_HD_in) == (48, 30, 8, None, True)
    with pytest.raises(KeyError):
        get_config('p', 'm', 'config_nope')


def test_config_modules_importable_like_reference():
    import importlib
    for name in CONFIG_NAMES:
        m = importlib.import_module('refvsr_amd.configs.' + name)
        assert m.get_config('p', 'm', name).network == ('RefVSR_IR' if '_IR_' in name else 'RefVSR')


def test_param_counts():
    want = {'config_RefVSR_small_L1': 2492070, 'config_RefVSR_MFID': 5717550,
            'config_RefVSR_MFID_8K': 5883185, 'config_RefVSR_small_MFID_8K': 2657705}
    for k, v in want.items():
        assert weights.num_params(get_config('p', 'm', k)) == v


def test_state_dict_generator_is_deterministic(small_cfg):
    a, b = make_state_dict(small_cfg, 1234), make_state_dict(small_cfg, 1234)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = make_state_dict(small_cfg, 1)
    assert not torch.equal(a['Network.conv_hr.weight'], c['Network.conv_hr.weight'])
    w = a['Network.feature_match.sub_mean.weight']
    assert torch.allclose(w[:, :, 0, 0], torch.diag(1.0 / torch.tensor(weights.VGG_STD)))


def _slot_enumeration(ks, ncg):
    """The K order by construction (independent of the closed form in packing.kslot / common.h:rv_kslot): even-parity
    blocks in natural order, one zero block if their count is odd, the odd-parity blocks, zero blocks up to 4*S."""
    nat = [(ty, tx, cg) for ty in range(ks) for tx in range(ks) for cg in range(ncg)]
    ev = [b for b in nat if (b[1] + b[2]) % 2 == 0]
    od = [b for b in nat if (b[1] + b[2]) % 2 == 1]
    order = ev + ([None] if len(ev) % 2 else []) + od
    return order + [None] * (-len(order) % 4)


@pytest.mark.parametrize('ks', [1, 3, 5, 7])
def test_kblock_order_closed_form(ks):
    from refvsr_amd.packing import keven, kslot, ksteps
    for ncg in range(1, 20):
        order = _slot_enumeration(ks, ncg)
        assert ksteps(ks, ncg) * 4 == len(order)
        assert keven(ks, ncg) == sum(1 for b in order if b is not None and (b[1] + b[2]) % 2 == 0)
        for j, b in enumerate(order):
            if b is not None:
                assert kslot(b[0], b[1], b[2], ks, ncg) == j
        # every ds_read_b128 lane group mixes slots (4s, 4s+1) or (4s+2, 4s+3): same parity, or one of them is a zero block
        for j in range(0, len(order), 2):
            a, b = order[j], order[j + 1]
            if a is not None and b is not None:
                assert (a[1] + a[2]) % 2 == (b[1] + b[2]) % 2


def _emulate(srcs, pk, ks, stride, pad, ho, wo):
    """numpy model of conv_mfma.hip's K walk: K-block g -> (tap, channel group) -> strided gather."""
    X = np.concatenate(srcs, -1)
    H, W, Ct = X.shape
    ncg = Ct // (4 if pk['f32'] else 8)
    Xp = np.zeros((H + 2 * pad + 2 * stride + ks, W + 2 * pad + 2 * stride + ks, Ct), np.float32)
    Xp[pad:pad + H, pad:pad + W] = X
    wp = pk['wpack'].float().numpy()
    if not pk['f32']:
        wp = wp.sum(3)                         # fp16 hi + lo
    nz, S, MT = wp.shape[:3]
    grp = 4 if pk['f32'] else 8
    out = np.zeros((nz * MT * 16, ho, wo), np.float32)
    order = _slot_enumeration(ks, ncg)
    assert len(order) == 4 * S
    for z in range(nz):
        for m in range(MT):
            for s in range(S):
                for q in range(4):
                    if order[4 * s + q] is None:
                        assert not wp[z, s, m, q * 16:(q + 1) * 16].any()     # padded K-blocks carry zero weights
                        continue
                    ty, tx, cg = order[4 * s + q]
                    A = wp[z, s, m, q * 16:(q + 1) * 16]
                    B = Xp[ty:ty + ho * stride:stride, tx:tx + wo * stride:stride, cg * grp:cg * grp + grp]
                    out[(z * MT + m) * 16:(z * MT + m + 1) * 16] += np.einsum('rk,yxk->ryx', A, B)
    return out + pk['bias'].numpy()[:, None, None]


@pytest.mark.parametrize('co,cins,ks,stride,shuffle,f32', [
    (24, [24], 3, 1, False, False), (24, [3, 24], 3, 1, False, False), (96, [24], 3, 1, True, False),
    (32, [32, 32], 5, 2, False, False), (2, [16], 7, 1, False, False), (48, [48, 48], 1, 1, False, False),
    (64, [32], 7, 1, False, False), (3, [24], 3, 1, False, False), (192, [48], 3, 1, True, False),
    (64, [3], 3, 1, False, True), (64, [64], 3, 1, False, True), (16, [64], 1, 1, False, True)])
def test_weight_packing_reproduces_conv(co, cins, ks, stride, shuffle, f32):
    rs = np.random.RandomState(0)
    cin = sum(cins)
    w = (rs.randn(co, cin, ks, ks) * 0.1).astype(np.float32)
    b = rs.randn(co).astype(np.float32)
    H, W, pad = 9, 11, ks // 2
    x = rs.randn(1, cin, H, W).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=stride, padding=pad)[0].numpy()
    pk = pack_conv(w, b, cins, shuffle, f32=f32)
    if f32:
        assert pk['wpack'].dtype == torch.float32 and pk['wpack'].shape[3:] == (64, 4)
    else:                                       # [nz, S, MT, hi|lo, lane, 8]
        assert pk['wpack'].dtype == torch.float16 and pk['wpack'].shape[3:] == (2, 64, 8)
    assert pk['wpack'].shape[0] * pk['mt'] * 16 >= co and pk['mt'] == choose_mt(co)
    grp = 4 if f32 else 8
    srcs, o = [], 0
    for c in cins:
        a = np.zeros((H, W, (c + grp - 1) // grp * grp), np.float32)
        a[:, :, :c] = x[0, o:o + c].transpose(1, 2, 0)
        srcs.append(a)
        o += c
    out = _emulate(srcs, pk, ks, stride, pad, ref.shape[1], ref.shape[2])
    if shuffle:
        C = co // 4
        rows = (np.arange(co) % C) * 4 + np.arange(co) // C
        got = np.zeros_like(ref)
        got[rows] = out[:co]
    else:
        got = out[:co]
    assert np.abs(got - ref).max() < 2e-5          # hi+lo fp16 weights carry ~22 bits; f32 mode is exact


def test_weight_packing_hi_only():
    """Weight mode 2 (RefvsrConv.f32 = 2, SPyNet's streamed 7x7 convs): ONE fp16 fragment per (K-step, m-tile), equal to the hi
    fragment of the hi + lo packing; only for streamed shapes (more than 16 K-steps)."""
    rs = np.random.RandomState(1)
    w = (rs.randn(64, 32, 7, 7) * 0.05).astype(np.float32)
    b = rs.randn(64).astype(np.float32)
    full = pack_conv(w, b, [32])
    for mt in (None, 1):
        hi = pack_conv(w, b, [32], mt=mt, hi_only=True)
        ref = pack_conv(w, b, [32], mt=mt)
        assert hi['hi_only'] and not hi['f32'] and hi['wpack'].dtype == torch.float16
        assert hi['wpack'].shape == ref['wpack'].shape[:3] + (1, 64, 8) and hi['ksteps'] == full['ksteps'] == 49
        assert torch.equal(hi['wpack'][:, :, :, 0], ref['wpack'][:, :, :, 0]) and torch.equal(hi['bias'], ref['bias'])
    with pytest.raises(AssertionError):
        pack_conv((rs.randn(24, 24, 3, 3) * 0.1).astype(np.float32), b[:24], [24], hi_only=True)     # resident shape: 7 K-steps


@pytest.mark.parametrize('c', [24, 48])
def test_conv_shuffle2_row_groups(c):
    """packing.pack_conv_shuffle2 (refvsr_conv_shuffle2): the row groups of the C -> 4 C conv, each packed like a 48-output conv,
    scattered by the kernel's rule (C = 24: blob z = dy, row R -> dx = R // 24, channel R % 24; C = 48: blob z = 2 dy + dx, row R =
    channel R) give F.pixel_shuffle(conv, 2); blob sizes as the library reports them."""
    from refvsr_amd import hip
    from refvsr_amd.packing import pack_conv24, pack_conv_shuffle2
    g = torch.Generator().manual_seed(c)
    w = torch.randn(4 * c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    b = torch.randn(4 * c, generator=g) * 0.1
    x = torch.randn(1, c, 6, 7, generator=g)
    want = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2)[0]
    blobs = pack_conv_shuffle2(w, b)
    nz = 2 if c == 24 else 4
    assert hip.lib().refvsr_conv_shuffle2_supported(c) == 1 and hip.lib().refvsr_conv_shuffle2_supported(36) == 0
    assert blobs.numel() == hip.lib().refvsr_conv_shuffle2_blob_bytes(c) and blobs.numel() % nz == 0
    got = torch.zeros_like(want)
    per = blobs.numel() // nz
    for z in range(nz):
        R = np.arange(48)
        rows = 4 * (R % 24) + 2 * z + R // 24 if c == 24 else 4 * R + z
        assert torch.equal(blobs[z * per:(z + 1) * per], pack_conv24(w[rows], b[rows], [c], shuffle_group=True))
        y = F.conv2d(x, w[rows], b[rows], padding=1)[0]                      # what the kernel's accumulators of group z hold
        for r in range(48):
            dy, dx, ch = (z, r // 24, r % 24) if c == 24 else (z >> 1, z & 1, r)
            got[ch, dy::2, dx::2] = y[r]
    assert torch.equal(got, want)


def test_engine_weight_modes(small_cfg, small_sd):
    """Which convs carry which weight representation (refvsr_amd/engine.py:Weights): SPyNet's streamed 7x7 convs (every conv of
    a level but the first, and their mt = 1 variants) plain fp16 (descriptor mode 2), everything else fp16 hi + lo, the VGG
    head of the matching exact fp32; config.spynet_hi_lo restores hi + lo in SPyNet."""
    import copy
    from refvsr_amd.engine import Weights
    W = Weights(small_cfg, small_sd, torch.device('cpu'))
    modes = collections.Counter()
    for name, cw in W.conv.items():
        spy = name.startswith('FlowNet.') and '.basic_module.0.conv' not in name.split('FlowNet.basic_module.')[1][1:]
        assert cw.hi_only == spy, name
        assert cw.desc.f32 == (2 if spy else (1 if cw.f32 else 0)), name
        assert cw.wpack.shape[3] == (1 if spy else 2) or cw.f32, name
        modes[cw.desc.f32] += 1
    assert modes[2] == 6 * 4 + 6 * 2 and modes[1] == 3 and modes[0] > 100          # 6 levels x (4 convs + 2 mt = 1 variants)
    cfg2 = copy.deepcopy(small_cfg)
    cfg2.spynet_hi_lo = True
    W2 = Weights(cfg2, small_sd, torch.device('cpu'))
    assert not any(cw.hi_only for cw in W2.conv.values())


def test_model_shell_state_dict_contract(small_cfg, small_sd):
    from refvsr_amd import SRNet
    net = SRNet(small_cfg)
    assert list(net.state_dict().keys()) == list(small_sd.keys())
    net.load_state_dict(small_sd, strict=True)
    net.load_state_dict(collections.OrderedDict(('module.' + k, v) for k, v in small_sd.items()), strict=True)
    assert torch.equal(net.state_dict()['Network.conv_last.bias'], small_sd['Network.conv_last.bias'])
    assert hasattr(net.Network, 'FlowNet') and hasattr(net.Network.FlowNet, 'load_ckpt')
    net.init()
    # plugin hook: unknown arch -> ImportError like the reference's importlib lookup
    bad = get_config('p', 'm', 'config_RefVSR_small_L1')
    bad.network = 'NoSuchArch'
    with pytest.raises(ImportError):
        SRNet(bad)


def test_product_path_has_no_cpu_fallback(small_cfg, small_sd):
    from refvsr_amd import SRNet
    net = SRNet(small_cfg)
    net.load_state_dict(small_sd)
    x = torch.rand(1, 5, 3, 16, 16)
    with pytest.raises(RuntimeError, match='GPU only'):
        net(x, x, True)
    with pytest.raises(NotImplementedError):
        net(x, x, True, is_train=True)


def test_product_never_imports_oracle():
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'refvsr_amd')
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f


def test_window_indices_and_clip():
    assert window_indices(0, 10, 5) == [0, 0, 0, 1, 2]
    assert window_indices(9, 10, 5) == [7, 8, 9, 9, 9]
    assert window_indices(4, 10, 3) == [3, 4, 5]
    lr, rf, gt = make_clip(3, 16, 24, seed=0)
    assert lr.shape == (3, 3, 16, 24) and rf.shape == lr.shape and gt.shape == (3, 3, 64, 96)
    lr2, _, _ = make_clip(2, 16, 24, seed=0, start=1)
    assert torch.equal(lr[1:], lr2)                       # shards of one endless clip line up
    assert float(lr.min()) >= 0 and float(lr.max()) <= 1
    assert torch.equal(torch.round(lr * 255) / 255, lr)   # 8-bit quantised
    assert not torch.equal(lr[0], lr[1])


def test_partition():
    assert shard.partition(64, 8) == [(8 * r, 8 * r + 8) for r in range(8)]
    assert shard.partition(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    p = shard.partition(64, 8, reset_branch=9, aligned=True)            # 8 units {0-8,...,63}
    assert p == [(0, 9), (9, 18), (18, 27), (27, 36), (36, 45), (45, 54), (54, 63), (63, 64)]
    p = shard.partition(64, 2, reset_branch=9, aligned=True)
    assert p == [(0, 36), (36, 64)] and all(s % 9 == 0 for s, _ in p)
    assert not shard.needs_handoff(0, None) and shard.needs_handoff(8, None)
    assert not shard.needs_handoff(18, 9) and shard.needs_handoff(20, 9)


def test_wavefront_partitions_and_makespan_model():
    """shard.partition_hybrid / partition_chain / predicted_speedup: the schedules bench.py uses for BASELINE configs[3] / [4]."""
    parts = shard.partition_hybrid(64, 8, 9)
    assert parts == [(0, 9), (9, 18), (18, 27), (27, 36), (36, 45), (45, 54), (54, 59), (59, 64)]
    assert sum(shard.needs_handoff(a, 9) for a, _ in parts) == 1
    assert shard.partition_hybrid(18, 2, 9) == [(0, 9), (9, 18)]                      # nothing to re-balance
    ch = shard.partition_chain(64, 8, 0.165)
    assert [b - a for a, b in ch] == [4, 5, 6, 7, 8, 10, 11, 13] and ch[0][0] == 0 and ch[-1][1] == 64
    assert all(b0 == a1 for (_, b0), (a1, _) in zip(ch, ch[1:]))
    assert shard.partition_chain(3, 8) == shard.partition(3, 8)
    ta, tb1, tb2 = 5.77, 0.946, 0.552                                                 # ms per frame, profiles/r03_bench.json
    s_h, _ = shard.predicted_speedup(64, 8, parts, 9, ta, tb1, tb2, 0.3)
    s_b, _ = shard.predicted_speedup(64, 8, shard.partition(64, 8), 9, ta, tb1, tb2, 0.3)
    s_n, _ = shard.predicted_speedup(64, 8, shard.partition(64, 8), None, ta, tb1, tb2, 0.3)
    s_c, _ = shard.predicted_speedup(64, 8, ch, None, ta, tb1, tb2, 0.3)
    assert abs(s_h - 64 / 9.0) < 0.03                    # exchange-free shards of <= 9 frames: 7.1 x
    assert 4.0 < s_b < 4.3 and abs(s_b - s_n) < 0.02     # a hand-off at every boundary: the B1 chain of all 64 frames
    assert s_c > 4.7 and s_c > s_n + 0.6                 # growing shards: the chain arrives when phase A ends (4.76 x)
    # round 4: block lists, the two-lane (interleaved) order, the cold window of a block start
    cyc = shard.partition_cyclic(64, 8, 3)
    assert cyc[:3] == [(0, 3, 0), (3, 6, 1), (6, 9, 2)] and cyc[-1] == (63, 64, 5) and len(cyc) == 22
    assert shard.as_blocks(ch) == [(a, b, r) for r, (a, b) in enumerate(ch)]
    cold = 4.9
    s_cyc, _ = shard.predicted_speedup(64, 8, cyc, None, ta, tb1, tb2, 0.3, cold)
    s_cyc_old, _ = shard.predicted_speedup(64, 8, cyc, None, ta, tb1, tb2, 0.3, cold, interleaved=False)
    s_grow_cold, _ = shard.predicted_speedup(64, 8, ch, None, ta, tb1, tb2, 0.3, cold)
    assert 5.2 < s_cyc < 5.5 and s_cyc_old < 3.7         # B1(f) as soon as ITS phase A is done: what makes small blocks pay
    assert 4.4 < s_grow_cold < 4.6                       # the best contiguous partition under the same cost model
    blocks, sp_, name = shard.choose_partition(64, 8, None, ta, tb1, tb2, 0.3, cold)
    assert name == 'block_cyclic_3' and abs(sp_ - s_cyc) < 1e-9 and blocks == shard.as_blocks(cyc)
    blocks, sp_, name = shard.choose_partition(64, 8, 9, ta, tb1, tb2, 0.3, cold)
    # (7.1 x without the cold block starts; a restart-aligned block start prepares FOUR extra contexts: two cold terms)
    assert name == 'hybrid_reset_aligned' and 6.0 < sp_ < 6.4
    # every partition, every order: each frame's three phases run exactly once (the simulation terminates)
    for parts_ in (shard.partition(13, 4), shard.partition_cyclic(13, 4, 2), shard.partition_chain(13, 4)):
        for rb in (None, 4):
            for il in (True, False):
                sp2, span = shard.predicted_speedup(13, 4, parts_, rb, 2.0, 0.5, 0.25, 0.1, 1.0, il)
                assert 1.0 <= sp2 <= 4.0 + 1e-9 and span >= 13 * 2.75 / 4 - 1e-9
    one, _ = shard.predicted_speedup(64, 1, [(0, 64)], 9, ta, tb1, tb2)
    assert abs(one - 1.0) < 2e-3                         # (time-stepped simulation: 1 / 2000 of a frame per step)


def test_block_chain_applies_blocks_in_order(monkeypatch):
    """Engine._block_chain: one fused launch per block (or two conv launches when fusing is off), in order."""
    from refvsr_amd import engine, ops

    class X(object):                               # stands in for an nhwc16 map: records what was applied to it
        def __init__(self, shape, hist=()):
            self.shape, self.hist = shape, tuple(hist)

    monkeypatch.setattr(ops, 'resblock', lambda c1, c2, x, act, post=1.0: X(x.shape, x.hist + ((c1, c2, act),)))
    monkeypatch.setattr(ops, 'conv', lambda cw, x, act=1.0, res=None: X(x.shape, x.hist + ((cw, act, res is not None),)))

    class E(object):
        _block_chain = engine.Engine._block_chain
        fuse_resblocks = True
        rb24 = False                             # the generic kernels (the 24-channel kernel is dispatched below)
        rb48, rb48_max_pixels = False, 0         # (the fused 48-channel block has its own dispatch test on the GPU)
        chain_events = None
        chain_calls = False                      # per-block launches from Python
    e = E()
    for n in (1, 2, 5, 24):
        pairs = [('a%d' % i, 'b%d' % i) for i in range(n)]
        assert e._block_chain(X((270, 480, 24)), pairs, 0.2).hist == tuple((a, b, 0.2) for a, b in pairs)
    # chain_calls: the whole run goes to ONE library call, with a pointer table cached on the packed weights
    calls = []
    monkeypatch.setattr(ops, 'resblock_chain_ok', lambda c: True)
    monkeypatch.setattr(ops, 'ResblockChain', lambda pairs: ('chain', tuple(pairs)))
    monkeypatch.setattr(ops, 'resblock_chain', lambda ch, x, act, post=1.0: calls.append((ch, act)) or X(x.shape, x.hist + (ch,)))
    e.chain_calls = True
    e.W = type('W', (), {'chains': {}})()
    out = e._block_chain(X((270, 480, 24)), pairs, 0.2)
    out2 = e._block_chain(X((270, 480, 24)), pairs, 0.2)
    assert out.hist == (('chain', tuple(pairs)),) and out2.hist == out.hist and len(calls) == 2 and len(e.W.chains) == 1
    # 24-channel maps whose packed weights kept their fp32 originals go to the specialised kernel: one blob table per run
    CW = collections.namedtuple('CW', 'name raw')
    pairs24 = [(CW('a%d' % i, ('w', 'b')), CW('b%d' % i, ('w', 'b'))) for i in range(3)]
    monkeypatch.setattr(ops, 'Resblock24Chain', lambda pairs, dev: ('rb24', tuple(pairs)))
    monkeypatch.setattr(ops, 'resblock24_chain', lambda ch, x, act: X(x.shape, x.hist + (ch, act)))
    e.rb24 = True
    x24 = X((270, 480, 24))
    x24.device = 'dev'
    assert e._block_chain(x24, pairs24, 0.0).hist == (('rb24', tuple(pairs24)), 0.0) and len(e.W.chains) == 2
    x48 = X((270, 480, 48))
    assert e._block_chain(x48, pairs, 0.2).hist[0][0] == 'chain'            # other channel counts: the generic chain call
    e.rb24 = False
    e.chain_calls = False
    e.fuse_resblocks = False
    want = tuple(x for a, b in pairs for x in ((a, 0.0, False), (b, 1.0, True)))
    assert e._block_chain(X((270, 480, 24)), pairs, 0.0).hist == want


def test_split_fp16_dot_product_is_fp32_grade():
    """The arithmetic of the exhaustive arg-max search (match.hip: match_patches lo rows + match_exact_kernel), restated in
    numpy: a = a_h + 2^-11 a_l with a_h = fp16(a), a_l = fp16((a - a_h) * 2^11); <a, b> ~ <a_h, b_h> + 2^-11 (<a_h, b_l> +
    <a_l, b_h>) with fp32 accumulation.  On normalised 144-element patches the result must be as close to the exact
    correlation as an fp32 dot product is (the claim DESIGN.md 4.1 makes), and far closer than the fp16 GEMM's score."""
    rs = np.random.RandomState(7)
    n, k = 4000, 144
    a = np.abs(rs.randn(n, k)).astype(np.float32)            # ReLU features are non-negative
    b = (a + 0.3 * np.abs(rs.randn(n, k))).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    exact = np.sum(a.astype(np.float64) * b.astype(np.float64), axis=1)

    def split(x):
        hi = x.astype(np.float16)
        lo = ((x - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
        return hi.astype(np.float32), lo.astype(np.float32)
    ah, al = split(a)
    bh, bl = split(b)
    assert np.abs((ah + al / 2048.0) - a).max() < 2.0 ** -22 * 1.01 * np.abs(a).max() + 1e-9      # representation: 22 bits
    assert np.all(np.abs(al) >= 0) and np.abs(al).max() <= 1.0 + 1e-3                              # lo * 2^11 stays a normal fp16
    f32 = lambda x: x.astype(np.float32)
    hh = np.zeros(n, np.float32)
    cross = np.zeros(n, np.float32)
    for j in range(k):                                        # fp32 accumulation, term by term
        hh = f32(hh + ah[:, j] * bh[:, j])
        cross = f32(cross + f32(ah[:, j] * bl[:, j]) + f32(al[:, j] * bh[:, j]))
    got = f32(hh + cross * np.float32(1.0 / 2048.0))
    fp32_dot = np.zeros(n, np.float32)
    for j in range(k):
        fp32_dot = f32(fp32_dot + a[:, j] * b[:, j])
    gemm16 = np.sum(ah.astype(np.float64) * bh.astype(np.float64), axis=1)                          # what match_top2 ranks by
    e_split, e_f32, e_16 = np.abs(got - exact).max(), np.abs(fp32_dot - exact).max(), np.abs(gemm16 - exact).max()
    # (term-by-term fp32 accumulation of 144 products near 1.0 is itself ~6e-7 off; the MFMA's blocked sums are tighter)
    assert e_split < 1e-6 and e_split < 1.5 * e_f32, (e_split, e_f32)
    assert e_16 > 20.0 * e_split, (e_16, e_split)              # the fp16 GEMM alone is two orders of magnitude coarser



# ---- the 24-channel fused block (csrc/resblock24.hip): lane- and address-level numpy model of one workgroup ------------------
def _rb24_model_tile(src, blob, h, w, ty0, tx0, relu, slope, nwv=8):
    """One 8 x 32 output tile as resblock24_kernel computes it: the SAME byte offsets into a 64 000-byte LDS image, the same
    K order, fragment rows, lane -> (row, pixel) maps and half-wave folds, written from the kernel's header comment and
    constants.  What it pins: the blob layout of packing.pack_resblock24 and the address algebra (window bases, K-step
    immediates, where t lands, where the residual is read, which lane stores which output channels).
    Returns (t_writes {(row, col, ch) of the x tile: value}, out {(oy, ox, ch): value})."""
    XW, IW, NI, PXB = 36, 34, 340, 48
    ROWB, WB, BIAS, XT = XW * PXB, 21504, 43008, 43264
    lds = np.zeros(64000, np.uint8)
    lds[:43264] = blob
    xt = np.zeros((12, XW, 24), np.float16)                       # x tile, zero padded
    for r in range(12):
        for c in range(XW):
            iy, ix = ty0 - 2 + r, tx0 - 2 + c
            if 0 <= iy < h and 0 <= ix < w:
                xt[r, c] = src[iy, ix]
    lds[XT:XT + xt.size * 2] = xt.reshape(-1).view(np.uint8)
    f16 = lambda off, n: lds[off:off + 2 * n].view(np.float16).astype(np.float32)
    f32 = lambda off, n: lds[off:off + 4 * n].view(np.float32).copy()
    pix16 = lambda n: ((n & 7) << 1) if (n < 4 or n >= 12) else (((n - 4) << 1) | 1)
    perm = (0, 2, 1, 3)
    lanes = range(64)

    def kloop(acc0, acc1, wofs, pb):           # acc*[lane] = 4 floats (rows 4q + i of column n); pb[lane] = window base
        for s in range(7):
            A = [np.stack([f16(wofs + (s * 3 + f) * 1024 + l * 16, 8) for l in lanes]) for f in range(3)]       # [64, 8]
            B = np.stack([f16(pb[l] + ((s >> 1) * ROWB + (s & 1) * 64 if s < 6
                                        else min(l >> 4, 2) * ROWB + 128 - perm[l >> 4] * 16), 8) for l in lanes])
            for f, acc in ((0, acc0), (2, acc1), (1, acc0)):
                D = np.zeros((16, 16), np.float32)                   # D[row][col] += A[row][k] B[k][col], k = 8 q + j
                for qq in range(4):
                    D += A[f][qq * 16:qq * 16 + 16] @ B[qq * 16:qq * 16 + 16].T
                for l in lanes:
                    for i in range(4):
                        acc[l][i] += D[4 * (l >> 4) + i, l & 15]

    fold = lambda acc: [[acc[l][i] + acc[l ^ 32][i] for i in range(4)] for l in lanes]
    act = (lambda v: max(v, 0.0)) if relu else (lambda v: max(v, v * slope))
    t1, rem, t2 = (22 + nwv - 1) // nwv, 22 % nwv, 16 // nwv
    # ---- phase 1 (+ the residual reads, which happen before barrier A)
    p1, xres = [], {}
    for wave in range(nwv):
        full = rem == 0 or wave < rem
        g1 = wave * t1 if full else rem * t1 + (wave - rem) * (t1 - 1)
        for t in range(t1 if full else t1 - 1):
            pb, pixs = [], []
            for l in lanes:
                pix = min((g1 + t) * 16 + pix16(l & 15), NI - 1)
                r = pix // IW
                pb.append(XT + r * ROWB + (pix - r * IW) * PXB + perm[l >> 4] * 16)
                pixs.append(pix)
            a0 = [list(f32(BIAS + (l >> 4) * 16, 4)) for l in lanes]
            a1 = [list(f32(BIAS + 64 + (l >> 4) * 16, 4)) for l in lanes]
            kloop(a0, a1, 0, pb)
            p1.append((pb, pixs, a0, fold(a1)))
        oy0 = (wave * t2) >> 1
        for t in range(t2):
            for l in lanes:
                q = l >> 4
                pb2 = XT + (oy0 + (t >> 1) + 1) * ROWB + ((t & 1) * 16 + pix16(l & 15) + 1) * PXB + perm[q] * 16
                dq = ROWB + PXB + q * 8 - perm[q] * 16
                xres[(wave, t, l)] = (pb2, f16(pb2 + dq, 4), f16(pb2 + dq + 32 if q < 2 else BIAS + 96, 4))
    # ---- barrier A; t over the x tile
    t_writes = {}
    for pb, pixs, a0, a1 in p1:
        for l in lanes:
            q = l >> 4
            r = pixs[l] // IW
            iy, ix = ty0 - 1 + r, tx0 - 1 + (pixs[l] - r * IW)
            keep = 0 <= iy < h and 0 <= ix < w
            for acc, extra in ((a0, 0), (a1, 32)):
                if extra and q >= 2:
                    continue
                v = np.array([act(np.float32(x)) if keep else 0.0 for x in acc[l]], np.float32).astype(np.float16)
                o = pb[l] + (ROWB + PXB + q * 8 - perm[q] * 16) + extra
                lds[o:o + 8] = v.view(np.uint8)
                rel = o - XT
                for i in range(4):
                    t_writes[(rel // ROWB, (rel % ROWB) // PXB, (rel % PXB) // 2 + i)] = float(v[i])
    # ---- barrier B; phase 2
    out = {}
    for wave in range(nwv):
        oy0 = (wave * t2) >> 1
        for t in range(t2):
            pb = [xres[(wave, t, l)][0] for l in lanes]
            c0 = [list(f32(BIAS + 128 + (l >> 4) * 16, 4) + xres[(wave, t, l)][1]) for l in lanes]
            c1 = [list(f32(BIAS + 192 + (l >> 4) * 16, 4) + xres[(wave, t, l)][2]) for l in lanes]
            kloop(c0, c1, WB, pb)
            c1 = fold(c1)
            for l in lanes:
                q = l >> 4
                oy, ox = ty0 + oy0 + (t >> 1), tx0 + (t & 1) * 16 + pix16(l & 15)
                if oy >= h or ox >= w:
                    continue
                for i in range(4):
                    out[(oy, ox, 4 * q + i)] = float(np.float16(c0[l][i]))
                    if q < 2:
                        out[(oy, ox, 16 + 4 * q + i)] = float(np.float16(c1[l][i]))
    return t_writes, out


@pytest.mark.parametrize('relu', [True, False])
def test_resblock24_blob_and_address_model(relu):
    """packing.pack_resblock24 + the address algebra of csrc/resblock24.hip reproduce a 24-channel residual block on tiles that
    touch every frame border (13 x 40 frame: partial last tile row, partial last tile column); the K-block table equals the
    library's."""
    from refvsr_amd import hip
    from refvsr_amd.packing import pack_resblock24, rb24_kblock
    lib = hip.lib()
    seen = set()
    for s in range(7):
        for q in range(4):
            kb, v = rb24_kblock(s, q), lib.refvsr_resblock24_kblock(s, q)
            assert v == (-1 if kb is None else (kb[0] << 16 | kb[1] << 8 | kb[2]))
            seen.add(kb)
    assert len(seen - {None}) == 27 and lib.refvsr_resblock24_kblock(7, 0) == -2
    g = torch.Generator().manual_seed(5)
    C, h, w = 24, 13, 40
    w1 = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    w2 = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    blob = pack_resblock24(w1, b1, w2, b2).numpy()
    assert blob.shape == (hip.RESBLOCK24_BLOB_BYTES,)
    x = torch.randn(C, h, w, generator=g).half()
    src = x.permute(1, 2, 0).contiguous().numpy()                   # HWC fp16
    slope = 0.0 if relu else 0.2
    t_ref = F.leaky_relu(F.conv2d(x.float()[None], w1, b1, padding=1), slope)
    o_ref = (x.float() + F.conv2d(t_ref.half().float(), w2, b2, padding=1)[0])
    t_ref = t_ref[0]
    covered = set()
    for ty0, tx0 in ((0, 0), (8, 32), (8, 0)):
        tw, out = _rb24_model_tile(src, blob, h, w, ty0, tx0, relu, slope)
        assert len(tw) == 10 * 34 * 24                               # the whole 10 x 34 intermediate, every channel once
        worst = 0.0
        for (r, c, ch), v in tw.items():
            assert 1 <= r <= 10 and 1 <= c <= 34
            iy, ix = ty0 - 2 + r, tx0 - 2 + c                        # x-tile coordinates -> frame
            want = float(t_ref[ch, iy, ix]) if (0 <= iy < h and 0 <= ix < w) else 0.0
            worst = max(worst, abs(v - want))
        assert worst < 4e-3, worst                                   # fp16 rounding of t (|t| <~ 4)
        for (oy, ox, ch), v in out.items():
            assert abs(v - float(o_ref[ch, oy, ox])) < 8e-3, (oy, ox, ch, v, float(o_ref[ch, oy, ox]))
            covered.add((oy, ox, ch))
    assert len(covered) == (13 * 32 + 5 * 8) * 24                    # tiles (0,0), (1,0) whole, (1,1): 5 rows x 8 columns


@pytest.mark.parametrize('srcs,cout', [([24], 24), ([16], 24), ([3, 24], 24), ([24, 24], 24), ([48], 48), ([16], 48), ([3, 48], 48), ([32], 32), ([3], 32)])
def test_conv24_blob_reproduces_conv(srcs, cout):
    """packing.pack_conv24: the K-block table equals the library's (csrc/conv24.hip:c24_kblock, whose plan -- every block once,
    one immediate per step and pattern, equal slot parity inside a ds_read_b128 lane group -- is proven by a static_assert at
    compile time), and the blob's fragments contracted with the staged window at the plan's slot offsets give the convolution
    (hi + lo rows, half-wave fold of the third fragment) at every pixel of a frame with borders."""
    from refvsr_amd import hip
    from refvsr_amd.packing import c24_kblock, c24_steps, pack_conv24, _pad8
    lib = hip.lib()
    pads = [_pad8(c) for c in srcs]
    ncg = sum(pads) // 8
    S = c24_steps(ncg)
    nf, nb = {24: (3, 32), 32: (4, 32), 48: (6, 64)}[cout]
    sup, nbytes = {24: (lib.refvsr_conv24_supported, lib.refvsr_conv24_blob_bytes), 32: (lib.refvsr_conv32_supported, lib.refvsr_conv32_blob_bytes),
                   48: (lib.refvsr_conv48_supported, lib.refvsr_conv48_blob_bytes)}[cout]
    assert sup(pads[0], pads[1] if len(pads) > 1 else 0) == 1
    assert nbytes(pads[0], pads[1] if len(pads) > 1 else 0) == S * nf * 1024 + nb * 4
    for s in range(S):
        for q in range(4):
            kb, v = c24_kblock(ncg, s, q), lib.refvsr_conv24_kblock(ncg, s, q)
            assert v == (-1 if kb is None else (kb[0] << 16 | kb[1] << 8 | kb[2])), (ncg, s, q)
    assert lib.refvsr_conv24_kblock(ncg, S, 0) == -2 and lib.refvsr_conv24_supported(48, 0) == 0 and lib.refvsr_conv48_supported(24, 24) == 0 and lib.refvsr_conv32_supported(24, 0) == 0
    g = torch.Generator().manual_seed(ncg)
    cin = sum(srcs)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    blob = pack_conv24(w, b, srcs).numpy()
    frag = blob[:S * nf * 1024].view(np.float16).astype(np.float32).reshape(S, nf, 4, 16, 8)   # [s][f][q][row][8]
    bias = blob[S * nf * 1024:].view(np.float32)
    h_, w_ = 5, 7
    x = torch.randn(cin, h_, w_, generator=g).half().float()
    # staged window memory: padded channel groups per pixel, zero border
    xp = np.zeros((h_ + 2, w_ + 2, ncg, 8), np.float32)
    o = 0
    cgo = 0
    for c, pc in zip(srcs, pads):
        blk = np.zeros((h_, w_, pc), np.float32)
        blk[:, :, :c] = x[o:o + c].permute(1, 2, 0).numpy()
        xp[1:-1, 1:-1, cgo:cgo + pc // 8] = blk.reshape(h_, w_, pc // 8, 8)
        o += c
        cgo += pc // 8
    want = F.conv2d(x[None], w, b, padding=1)[0].numpy()
    got = np.zeros((cout, h_, w_), np.float32)
    for oy in range(h_):
        for ox in range(w_):
            acc = [bias[16 * m:16 * m + 16].copy() for m in range(nb // 16)]       # rows = channels 16 m .. (pads: zeros)
            if cout == 24:
                acc[1][8:] = 0.0                                     # rows 8-15 of the third fragment collect lo sums only
            for s in range(S):
                for q in range(4):
                    kb = c24_kblock(ncg, s, q)
                    if kb is None:
                        assert not frag[s, :, q].any()               # zero block: zero weights, whatever it reads
                        continue
                    ty, tx, cg = kb
                    bvec = xp[oy + ty, ox + tx, cg]
                    if cout == 24:
                        acc[0] += frag[s, 0, q] @ bvec + frag[s, 1, q] @ bvec
                        acc[1] += frag[s, 2, q] @ bvec
                    else:
                        for m in range(cout // 16):
                            acc[m] += frag[s, 2 * m, q] @ bvec + frag[s, 2 * m + 1, q] @ bvec
            if cout == 24:
                got[0:16, oy, ox] = acc[0]
                got[16:24, oy, ox] = acc[1][0:8] + acc[1][8:16]
            else:
                for m in range(cout // 16):
                    got[16 * m:16 * m + 16, oy, ox] = acc[m]
    assert np.abs(got - want).max() < 2e-5


def test_conv24_ok_equals_the_library_predicates():
    """packing.conv24_ok (which decides whether ConvWeights builds a specialised blob) must accept exactly the shapes
    refvsr_conv{24,32,48}_supported accept -- a plain 24 -> 48 conv is NOT one of them (only the row groups of the pixel-shuffle
    conv are packed that way, internally)."""
    from refvsr_amd import hip
    from refvsr_amd.packing import conv24_ok
    lib = hip.lib()
    sup = {24: lib.refvsr_conv24_supported, 32: lib.refvsr_conv32_supported, 48: lib.refvsr_conv48_supported}
    for cout in (16, 24, 32, 48, 64):
        for c0 in (3, 8, 16, 24, 32, 36, 48, 64):
            for c1 in (0, 8, 24, 48):
                srcs = [c0] + ([c1] if c1 else [])
                p0 = (c0 + 7) // 8 * 8
                want = bool(cout in sup and sup[cout](p0, c1))
                assert conv24_ok((cout, c0 + c1, 3, 3), srcs) == want, (cout, srcs)
    assert not conv24_ok((48, 24, 3, 3), [24]) and conv24_ok((48, 24, 3, 3), [24], shuffle_group=True)
    assert not conv24_ok((24, 24, 5, 5), [24]) and not conv24_ok((24, 24, 3, 3), [24], f32=True)


def test_partition_chain_gives_every_rank_a_frame():
    """no empty shards for nframes >= world, sizes non-decreasing along the chain, every frame exactly once."""
    from refvsr_amd import shard
    for nframes in (4, 6, 8, 9, 12, 13, 20, 64, 100):
        for world in (2, 4, 6, 8, 12):
            for ratio in (0.05, 0.165, 0.5, 1.0, 2.0):
                parts = shard.partition_chain(nframes, world, ratio)
                sizes = [b - a for a, b in parts]
                assert parts[0][0] == 0 and parts[-1][1] == nframes and all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
                if nframes >= world:
                    assert min(sizes) >= 1, (nframes, world, ratio, sizes)
                    assert sizes == sorted(sizes), (nframes, world, ratio, sizes)
    assert [b - a for a, b in shard.partition_chain(64, 8, 0.165)] == [4, 5, 6, 7, 8, 10, 11, 13]


def test_env_flag_parsing(monkeypatch):
    """NAME=0 switches a knob off (string truthiness used to switch it ON)."""
    from refvsr_amd.knobs import env_flag
    monkeypatch.delenv('REFVSR_TEST_KNOB', raising=False)
    assert env_flag('REFVSR_TEST_KNOB') is False and env_flag('REFVSR_TEST_KNOB', True) is True
    for v, want in (('1', True), ('0', False), ('', False), ('false', False), ('off', False), ('yes', True)):
        monkeypatch.setenv('REFVSR_TEST_KNOB', v)
        assert env_flag('REFVSR_TEST_KNOB') is want, v


def test_round4_blobs_follow_the_library_k_plan():
    """pack_resblock48 / pack_conv_last (host side of refvsr_resblock48_chain / refvsr_conv_last): sizes equal the library's
    constants, the 48-channel block blob is the two refvsr_conv48 blobs' fragment parts + both biases, and the output head's
    single fragment per K-step holds hi(W) in rows 0-2 and lo(W) in rows 8-10 of the K-block the library's plan
    (refvsr_conv24_kblock) assigns to (K-step, lane quarter) -- decoded back, hi + lo reproduces the fp32 weights to 2^-21."""
    from refvsr_amd import hip
    from refvsr_amd.packing import RB48_BLOB, RB48_WB, pack_conv24, pack_conv_last, pack_resblock48
    lib = hip.lib()
    g = torch.Generator().manual_seed(4)
    w1, w2 = torch.randn(48, 48, 3, 3, generator=g), torch.randn(48, 48, 3, 3, generator=g)
    b1, b2 = torch.randn(48, generator=g), torch.randn(48, generator=g)
    blob = pack_resblock48(w1, b1, w2, b2)
    assert blob.numel() == RB48_BLOB == hip.RESBLOCK48_BLOB_BYTES == 2 * lib.refvsr_conv48_blob_bytes(48, 0)
    c1, c2 = pack_conv24(w1, b1, [48]), pack_conv24(w2, b2, [48])
    assert torch.equal(blob[:RB48_WB], c1[:RB48_WB]) and torch.equal(blob[RB48_WB:2 * RB48_WB], c2[:RB48_WB])
    assert torch.equal(blob[2 * RB48_WB:2 * RB48_WB + 256], c1[RB48_WB:]) and torch.equal(blob[2 * RB48_WB + 256:], c2[RB48_WB:])
    for c in (24, 48):
        w = torch.randn(3, c, 3, 3, generator=g) * 0.1
        b = torch.randn(3, generator=g)
        hb = pack_conv_last(w, b).numpy()
        assert hb.size == lib.refvsr_conv_last_blob_bytes(c) and lib.refvsr_conv_last_supported(c) == 1
        ncg = c // 8
        S = (hb.size - 128) // 1024
        frag = hb[:S * 1024].view(np.float16).reshape(S, 4, 16, 8).astype(np.float32)      # [s][q][row][8 channels]
        assert np.allclose(hb[S * 1024:].view(np.float32)[:3], b.numpy()) and not hb[S * 1024:].view(np.float32)[3:].any()
        seen = np.zeros((3, c, 3, 3), np.float32)
        for s_ in range(S):
            for q in range(4):
                kb = lib.refvsr_conv24_kblock(ncg, s_, q)
                if kb < 0:
                    assert not frag[s_, q].any()                                           # zero block
                    continue
                ty, tx, cg = kb >> 16, (kb >> 8) & 255, kb & 255
                seen[:, cg * 8:cg * 8 + 8, ty, tx] += frag[s_, q, 0:3] + frag[s_, q, 8:11]
                assert not frag[s_, q, 3:8].any() and not frag[s_, q, 11:].any()
        assert np.abs(seen - w.numpy()).max() < 0.1 * 2.0 ** -20
    assert lib.refvsr_conv_last_supported(36) == 0 and lib.refvsr_conv_last_blob_bytes(36) == -1
    # refvsr_conv_hr_last: the block blob with conv1 = conv_hr and the head in conv2's fragment slot (s, 0)
    from refvsr_amd.packing import RB24_NF, RB24_S, pack_conv_hr_last, pack_resblock24, rb24_kblock
    w1, b1 = torch.randn(24, 24, 3, 3, generator=g), torch.randn(24, generator=g)
    w2, b2 = torch.randn(3, 24, 3, 3, generator=g) * 0.1, torch.randn(3, generator=g)
    hb = pack_conv_hr_last(w1, b1, w2, b2).numpy()
    blk = pack_resblock24(w1, b1, torch.zeros(24, 24, 3, 3), torch.zeros(24)).numpy()
    wb = RB24_S * RB24_NF * 1024
    assert hb.size == hip.RESBLOCK24_BLOB_BYTES and (hb[:wb] == blk[:wb]).all() and (hb[2 * wb:2 * wb + 128] == blk[2 * wb:2 * wb + 128]).all()
    f2 = hb[wb:2 * wb].view(np.float16).astype(np.float32).reshape(RB24_S, RB24_NF, 4, 16, 8)
    assert not f2[:, 1:].any() and not f2[:, 0, :, 3:8].any() and not f2[:, 0, :, 11:].any()
    seen = np.zeros((3, 24, 3, 3), np.float32)
    for s_ in range(RB24_S):
        for q in range(4):
            kb = rb24_kblock(s_, q)
            if kb is not None:
                seen[:, kb[2] * 8:kb[2] * 8 + 8, kb[0], kb[1]] += f2[s_, 0, q, 0:3] + f2[s_, 0, q, 8:11]
    assert np.abs(seen - w2.numpy()).max() < 0.1 * 2.0 ** -19
    assert np.allclose(hb[2 * wb + 128:].view(np.float32)[:3], b2.numpy()) and not hb[2 * wb + 128:].view(np.float32)[3:].any()


def test_conv48_two_source_blob_is_two_channel_half_blobs():
    """refvsr_conv48's 48 + 48 -> 48 form (the two-source convs of the mid_channels = 48 models): the blob is two 24-output blobs
    (channels 0-23 | 24-47) on the NCG = 12 K plan -- three K-steps per tap, channel groups 4 j + {0, 2, 1, 3}.  The python plan
    equals the library's, every (tap, channel group) occurs exactly once, and each half's fragments contracted with the staged
    window of the concatenated sources reproduce F.conv2d for that half (hi + lo rows, half-wave fold of the third fragment)."""
    from refvsr_amd import hip
    from refvsr_amd.packing import c24_kblock, c24_steps, pack_conv24
    lib = hip.lib()
    ncg, S = 12, c24_steps(12)
    assert S == 27 and lib.refvsr_conv48_supported(48, 48) == 1 and lib.refvsr_conv48_blob_bytes(48, 48) == 2 * (S * 3 * 1024 + 128)
    seen = set()
    for s in range(S):
        for q in range(4):
            kb = c24_kblock(ncg, s, q)
            assert kb is not None and lib.refvsr_conv24_kblock(ncg, s, q) == (kb[0] << 16 | kb[1] << 8 | kb[2])
            seen.add(kb)
    assert len(seen) == 9 * 12
    g = torch.Generator().manual_seed(12)
    w = torch.randn(48, 96, 3, 3, generator=g) / (96 * 9) ** 0.5
    b = torch.randn(48, generator=g) * 0.1
    blob = pack_conv24(w, b, [48, 48]).numpy()
    half = S * 3 * 1024 + 128
    assert blob.size == 2 * half
    h_, w_ = 3, 4
    x = torch.randn(96, h_, w_, generator=g).half().float()
    xp = np.zeros((h_ + 2, w_ + 2, ncg, 8), np.float32)
    xp[1:-1, 1:-1] = x.permute(1, 2, 0).numpy().reshape(h_, w_, ncg, 8)        # src0 groups 0-5, src1 groups 6-11
    want = F.conv2d(x[None], w, b, padding=1)[0].numpy()
    for z in range(2):
        hb = blob[z * half:(z + 1) * half]
        frag = hb[:S * 3 * 1024].view(np.float16).astype(np.float32).reshape(S, 3, 4, 16, 8)
        bias = hb[S * 3 * 1024:].view(np.float32)
        for oy in range(h_):
            for ox in range(w_):
                a0, a1 = bias[0:16].copy(), bias[16:32].copy()
                a1[8:] = 0.0
                for s in range(S):
                    for q in range(4):
                        ty, tx, cg = c24_kblock(ncg, s, q)
                        bvec = xp[oy + ty, ox + tx, cg]
                        a0 += frag[s, 0, q] @ bvec + frag[s, 1, q] @ bvec
                        a1 += frag[s, 2, q] @ bvec
                got = np.concatenate([a0, a1[:8] + a1[8:]])
                assert np.abs(got - want[24 * z:24 * z + 24, oy, ox]).max() < 2e-5, (z, oy, ox)


def _context_protocol_completes(plan, in_order_transport, group=1):
    """Replays ContextPlan.program of every rank.  in_order_transport=False: gloo -- the host blocks in wait_recv until the peer has
    ISSUED the matching send (messages of one direction of a pair match in posting order).  True: RCCL -- nothing blocks the host;
    lane a is an in-order stream [prep | send (the pair stream waits for lane a up to here) | wait_recv | a1], every rank pair has ONE
    in-order stream per side holding its sends and receives in issue order, and an operation runs only when it is at the head on
    BOTH sides.  Returns (completed, per-rank trace of 'a1' windows with the contexts available then)."""
    world, nfr = plan.world, plan.nframes
    from refvsr_amd import shard as _shard
    # group > 1: what run_wavefront(group=) executes -- runs of windows as one ('ag', members) op, everything between them issued first
    prog = {r: _shard.group_lane_ops(plan.program(r), group) for r in range(world)}
    members = lambda op: op[1] if op[0] == 'ag' else (op[1],)
    if not in_order_transport:
        pc = {r: 0 for r in range(world)}
        sent = {}                                  # (src, dst) -> list of frames in send order
        recv_posted = {}                           # (src, dst) -> list of frames in posting order at dst
        have = {r: set() for r in range(world)}
        progress = True
        while progress:
            progress = False
            for r in range(world):
                while pc[r] < len(prog[r]):
                    op = prog[r][pc[r]]
                    if op[0] == 'prep':
                        have[r].add(op[1])
                    elif op[0] == 'send':
                        assert op[1] in have[r]
                        sent.setdefault((r, op[2]), []).append(op[1])
                    elif op[0] == 'post_recv':
                        recv_posted.setdefault((op[2], r), []).append(op[1])
                    elif op[0] == 'wait_recv':
                        k = recv_posted[(op[2], r)].index(op[1])
                        s_ = sent.get((op[2], r), [])
                        if len(s_) <= k:
                            break                  # blocked: the peer has not issued that send yet
                        assert s_[k] == op[1], 'message order of the pair differs between its ends'
                        have[r].add(op[1])
                    else:
                        for f_ in members(op):
                            assert all(i in have[r] for i in plan.needed[f_]), 'window %d lacks a context' % f_
                    pc[r] += 1
                    progress = True
        return all(pc[r] == len(prog[r]) for r in range(world))
    # stream semantics
    lane = {r: [op for op in prog[r] if op[0] in ('prep', 'send', 'wait_recv', 'a1', 'ag')] for r in range(world)}
    if in_order_transport == 'one stream per rank':
        # eager-initialised ProcessGroupNCCL: unbatched send / recv of a group are serialised with ALL its other operations -- one
        # FIFO per rank over all peers; an operation runs when it and its match are at the heads of both FIFOs
        fifo = {r: [('send' if op[0] == 'send' else 'recv', op[1], op[2]) for op in prog[r] if op[0] in ('send', 'post_recv')]
                for r in range(world)}
        lp = {r: 0 for r in range(world)}
        fp = {r: 0 for r in range(world)}
        send_ready, received, have = set(), set(), {r: set() for r in range(world)}
        progress = True
        while progress:
            progress = False
            for r in range(world):
                while lp[r] < len(lane[r]):
                    op = lane[r][lp[r]]
                    if op[0] == 'prep':
                        have[r].add(op[1])
                    elif op[0] == 'send':
                        send_ready.add((r, op[2], op[1]))
                    elif op[0] == 'wait_recv':
                        if (r, op[1]) not in received:
                            break
                        have[r].add(op[1])
                    lp[r] += 1
                    progress = True
            for r in range(world):
                if fp[r] < len(fifo[r]):
                    k, i, peer = fifo[r][fp[r]]
                    if fp[peer] < len(fifo[peer]):
                        k2, i2, peer2 = fifo[peer][fp[peer]]
                        if peer2 == r and i2 == i and {k, k2} == {'send', 'recv'}:
                            src, dst = (r, peer) if k == 'send' else (peer, r)
                            if (src, dst, i) in send_ready:
                                received.add((dst, i))
                                fp[r] += 1
                                fp[peer] += 1
                                progress = True
        return all(lp[r] == len(lane[r]) for r in range(world)) and all(fp[r] == len(fifo[r]) for r in range(world))
    pair = {}                                      # (r, peer) -> [(kind, frame)] in issue order on r's side
    for r in range(world):
        for op in prog[r]:
            if op[0] in ('send', 'post_recv'):
                pair.setdefault((r, op[2]), []).append(('send' if op[0] == 'send' else 'recv', op[1]))
    lp = {r: 0 for r in range(world)}
    pp = {k: 0 for k in pair}
    send_ready, received, have = set(), set(), {r: set() for r in range(world)}
    progress = True
    while progress:
        progress = False
        for r in range(world):
            while lp[r] < len(lane[r]):
                op = lane[r][lp[r]]
                if op[0] == 'prep':
                    have[r].add(op[1])
                elif op[0] == 'send':
                    send_ready.add((r, op[2], op[1]))          # the pair stream may now run this send
                elif op[0] == 'wait_recv':
                    if (r, op[1]) not in received:
                        break
                    have[r].add(op[1])
                else:
                    for f_ in members(op):
                        assert all(i in have[r] for i in plan.needed[f_]), 'window %d lacks a context' % f_
                lp[r] += 1
                progress = True
        for (r, peer), ops_ in pair.items():
            if r > peer:
                continue
            a, b = ops_, pair.get((peer, r), [])
            while pp[(r, peer)] < len(a) and pp.get((peer, r), 0) < len(b):
                (ka, fa), (kb, fb) = a[pp[(r, peer)]], b[pp[(peer, r)]]
                assert fa == fb and {ka, kb} == {'send', 'recv'}, 'the heads of a pair do not match: %s / %s' % ((ka, fa), (kb, fb))
                src, dst = (r, peer) if ka == 'send' else (peer, r)
                if (src, dst, fa) not in send_ready:
                    break
                received.add((dst, fa))
                pp[(r, peer)] += 1
                pp[(peer, r)] += 1
                progress = True
    return all(lp[r] == len(lane[r]) for r in range(world)) and all(pp[k] == len(v) for k, v in pair.items())


def test_context_plan_covers_every_window_and_never_deadlocks():
    """ContextPlan (run_wavefront with exchange_contexts): every context prepared by exactly one rank, every window finds the
    contexts it needs, both ends of a rank pair issue their messages in the same order, and the programs complete under the
    blocking-host (gloo) and the in-order-stream (RCCL) semantics -- over worlds, clip lengths, restart periods, window lengths and
    every partition family, including ranks with several blocks and pairs with traffic in both directions."""
    from refvsr_amd import shard
    cases = serial_ok = 0
    for world in (2, 3, 4, 8):
        for nfr in (world, world + 1, 13, 26, 64):
            for reset in (None, 4, 9):
                for t in (3, 5, 7):
                    fams = [shard.partition(nfr, world), shard.partition_chain(nfr, world) if nfr >= world else None,
                            shard.partition_cyclic(nfr, world, 1), shard.partition_cyclic(nfr, world, 2), shard.partition_cyclic(nfr, world, 3),
                            shard.partition_cyclic_growing(nfr, world, 5.3, 1.04, 7.2), shard.partition_cyclic_growing(nfr, world, 5.3, 1.04, 7.2, 1.2)]
                    if reset:
                        fams.append(shard.partition_hybrid(nfr, world, reset))
                    for parts in fams:
                        if parts is None:
                            continue
                        plan = shard.ContextPlan(nfr, world, parts, reset, t)
                        preps = sorted(op[1] for r in range(world) for op in plan.program(r) if op[0] == 'prep')
                        assert preps == list(range(nfr)), (world, nfr, reset, t, parts)
                        for r in range(world):
                            for p in range(world):
                                if p != r:
                                    mine = [(i, k) for i, k in plan.pair_order(r, p)]
                                    theirs = [(i, 'send' if k == 'recv' else 'recv') for i, k in plan.pair_order(p, r)]
                                    assert mine == theirs
                        assert _context_protocol_completes(plan, False), ('gloo', world, nfr, reset, t, parts)
                        assert _context_protocol_completes(plan, True), ('rccl', world, nfr, reset, t, parts)
                        serial_ok += bool(_context_protocol_completes(plan, 'one stream per rank'))
                        cases += 1
    # (all context operations of a rank serialised on ONE stream -- what an eagerly initialised ProcessGroupNCCL does to unbatched
    #  send / recv; bench.py initialises lazily, one communicator and stream per rank pair -- must complete as well)
    assert cases > 1000 and serial_ok == cases


def test_grouped_lane_programs_keep_the_protocol():
    """run_wavefront(group=G): runs of up to G windows of a rank become one phase-A group (multi-map launches), everything that stood
    between them -- preparations, sends, receives -- is issued before the group.  The grouped programs hold every op of the plain ones,
    keep the order of the messages, find every context and complete under the three transport semantics."""
    from refvsr_amd import shard
    assert shard.group_lane_ops([('a1', 0), ('prep', 3), ('a1', 1), ('wait_recv', 4, 1), ('a1', 2), ('a1', 5), ('a1', 6)], 3) == \
        [('prep', 3), ('wait_recv', 4, 1), ('ag', (0, 1, 2)), ('ag', (5, 6))]
    assert shard.group_lane_ops([('a1', 0), ('a1', 1)], 1) == [('a1', 0), ('a1', 1)]
    assert shard.group_time(4, 4, 4.0, 5.5) == 4.0 and shard.group_time(1, 4, 4.0, 5.5) == 5.5 and shard.group_time(2, 4, 4.0, 5.5) == 5.0
    cases = 0
    for world in (2, 3, 8):
        for nfr in (world + 1, 13, 64):
            for reset in (None, 9):
                for t in (3, 5):
                    fams = [shard.partition(nfr, world), shard.partition_cyclic(nfr, world, 1), shard.partition_cyclic(nfr, world, 3),
                            shard.partition_cyclic_growing(nfr, world, 5.3, 1.04, 7.2)]
                    if reset:
                        fams.append(shard.partition_hybrid(nfr, world, reset))
                    for parts in fams:
                        plan = shard.ContextPlan(nfr, world, parts, reset, t)
                        for G in (2, 4):
                            for r in range(world):
                                plain, grouped = plan.program(r), shard.group_lane_ops(plan.program(r), G)
                                flat = [('a1', f) for op in grouped if op[0] == 'ag' for f in op[1]]
                                assert flat == [op for op in plain if op[0] == 'a1'] and all(len(op[1]) <= G for op in grouped if op[0] == 'ag')
                                assert [op for op in grouped if op[0] != 'ag'] == [op for op in plain if op[0] != 'a1']
                            assert _context_protocol_completes(plan, False, G), ('gloo', world, nfr, reset, t, G, parts)
                            assert _context_protocol_completes(plan, True, G), ('rccl', world, nfr, reset, t, G, parts)
                            assert _context_protocol_completes(plan, 'one stream per rank', G), ('serial', world, nfr, reset, t, G, parts)
                            cases += 1
    assert cases >= 300
    # the model: a group costs its members' time at the group rate and finishes them together; grouping never changes the work of a rank
    ta, ta1, tb1, tb2 = 4.2, 5.4, 0.9, 0.45
    hyb = shard.partition_hybrid(64, 8, 9)
    ex = dict(t_prep=2.2, t_ctx=0.1, t_cold_x=0.5)
    s1, span1 = shard.predicted_speedup(64, 8, hyb, 9, ta1, tb1, tb2, 0.1, 4.0, True, ex)
    s4, span4 = shard.predicted_speedup(64, 8, hyb, 9, ta, tb1, tb2, 0.1, 4.0, True, ex, group=4, t_a_single=ta1)
    assert span4 < span1 and span4 >= 9 * (ta + tb1 + tb2) - 1e-6           # never below the largest shard's own work at the group rate
    one = shard.predicted_speedup(64, 1, [(0, 64)], 9, ta, tb1, tb2, group=4, t_a_single=ta1)[0]
    assert abs(one - 1.0) < 0.02


def test_context_exchange_model():
    """simulate_wavefront with the context exchange, on the phase times measured in round 4 (ms): every context prepared once
    removes the cold block starts (two extra contexts per block, four at a restart).  Restart-free 64-frame clip on 8 ranks: the
    growing block-cyclic partition reaches the review's 5.5x; with restarts every 9 frames the reset-aligned partition reaches 7x
    (64 / 9 = 7.1 is its load-balance bound).  Without the exchange a restart-aligned block start costs TWO cold terms."""
    from refvsr_amd import shard
    ta, tb1, tb2, cold = 5.31, 0.89, 0.45, 3.77
    ex = dict(t_prep=1.9, t_ctx=0.3, t_cold_x=0.4)
    hyb = shard.partition_hybrid(64, 8, 9)
    s_no = shard.predicted_speedup(64, 8, hyb, 9, ta, tb1, tb2, 0.3, cold)[0]
    s_ex = shard.predicted_speedup(64, 8, hyb, 9, ta, tb1, tb2, 0.3, cold, True, ex)[0]
    assert 6.2 < s_no < 6.45 and 6.95 < s_ex < 7.12, (s_no, s_ex)
    blocks, sp, name = shard.choose_partition(64, 8, None, ta, tb1, tb2, 0.3, cold, exchange=ex)
    assert sp >= 5.5 and name.startswith('block_cyclic_growing'), (sp, name)
    sp0 = shard.choose_partition(64, 8, None, ta, tb1, tb2, 0.3, cold)[1]
    assert 5.25 < sp0 < sp
    # a free exchange of free contexts can only help a given partition
    cyc = shard.partition_cyclic(64, 8, 2)
    assert shard.predicted_speedup(64, 8, cyc, None, ta, tb1, tb2, 0.3, cold, True, dict(t_prep=1.9, t_ctx=0.0))[0] > \
        shard.predicted_speedup(64, 8, cyc, None, ta, tb1, tb2, 0.3, cold)[0]


def test_growing_block_cyclic_partition_and_choice_are_well_formed():
    """partition_cyclic_growing: contiguous cover of the clip, block sizes >= 1, non-decreasing up to the cap and the tail, owners
    dealt round-robin; choose_partition returns a valid block list for every (world, clip, restart) and never predicts more than
    the rank count or less than one; the exchange with free messages never loses to the same partition without it."""
    from refvsr_amd import shard
    ta, tb1, tb2, cold = 5.35, 0.89, 0.45, 3.9
    for world in (2, 3, 8):
        for nfr in (world + 1, 20, 64, 100):
            for scale in (0.8, 1.0, 1.2):
                blocks = shard.partition_cyclic_growing(nfr, world, ta, tb1 + 0.05, ta + 2.2, scale)
                assert blocks[0][0] == 0 and blocks[-1][1] == nfr and all(b0[1] == b1[0] for b0, b1 in zip(blocks, blocks[1:]))
                sizes = [b - a for a, b, _ in blocks]
                assert min(sizes) >= 1 and max(sizes) <= 8 and sizes[:-1] == sorted(sizes[:-1])
                assert [r for _, _, r in blocks] == [k % world for k in range(len(blocks))]
            for rb in (None, 9):
                for ex in (None, dict(t_prep=2.2, t_ctx=0.3, t_cold_x=0.0)):
                    blk, sp, name = shard.choose_partition(nfr, world, rb, ta, tb1, tb2, 0.1, cold, exchange=ex)
                    assert shard.as_blocks(blk, world) == blk and 1.0 - 1e-6 <= sp <= world + 1e-6, (world, nfr, rb, name, sp)
    cyc = shard.partition_cyclic(64, 8, 2)
    for rb in (None, 9):
        free = shard.predicted_speedup(64, 8, cyc, rb, ta, tb1, tb2, 0.1, cold, True, dict(t_prep=2.2, t_ctx=0.0, t_cold_x=0.0))[0]
        assert free >= shard.predicted_speedup(64, 8, cyc, rb, ta, tb1, tb2, 0.1, cold)[0] - 1e-6


def test_context_plan_leads_and_programs():
    """ContextPlan: the lead (how many chain steps ahead of a foreign window a context is prepared) only reorders a rank's tasks --
    every window still follows the contexts it needs, every context is prepared once, whatever the lead."""
    from refvsr_amd import shard
    for lead in (0, 1, 3, 8):
        for parts in (shard.partition_cyclic(26, 4, 2), shard.partition_hybrid(40, 4, 9), shard.partition_chain(26, 4)):
            rb = 9 if parts == shard.partition_hybrid(40, 4, 9) else None
            nfr = 40 if rb else 26
            plan = shard.ContextPlan(nfr, 4, parts, rb, 5, lead=lead)
            seen = []
            for r in range(4):
                have = set()
                for op in plan.program(r):
                    if op[0] == 'prep':
                        seen.append(op[1])
                        have.add(op[1])
                    elif op[0] == 'wait_recv':
                        have.add(op[1])
                    elif op[0] == 'a1':
                        assert set(plan.needed[op[1]]) <= have, (lead, r, op)
            assert sorted(seen) == list(range(nfr))
            assert _context_protocol_completes(plan, False) and _context_protocol_completes(plan, True)


def test_bench_wavefront_model_runs_on_cpu():
    """bench.wavefront_model (the N = 1 bench line's prediction object) is pure host code: it must run without a GPU, carry the
    context-exchange and two-message terms, and order its predictions sensibly."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pf = {'phase_a_ms': 5.35, 'phase_b1_ms': 0.89, 'phase_b2_ms': 0.45, 'phase_a_cold_extra_ms': 3.9, 'context_prepare_ms': 2.2,
          'phase_a_cold_with_contexts_ms': 2.9}
    ex = bench.exchange_terms(pf)
    assert abs(ex['t_prep'] - 2.2) < 1e-9 and ex['t_cold_x'] == 0.0 and ex['t_ctx'] == 0.3
    m = bench.wavefront_model(pf, 64, 9)
    assert abs(m['handoff_ms_on_the_chain'] - (0.24 * 0.3 + 0.03)) < 1e-9 and m['message_ms'] == 0.3
    e8 = m['predicted_speedup']['8']
    wr, nr = e8['with_restarts (reset_branch=9)'], e8['no_restarts (reset_branch=None, configs[4] regime)']
    assert 6.0 < wr['speedup'] < wr['with_context_exchange']['speedup'] <= 8.0
    assert 5.0 < nr['speedup'] < nr['with_context_exchange']['speedup'] < 6.5 and nr['with_context_exchange']['speedup'] >= 5.5
    assert sum(nr['with_context_exchange']['block_sizes']) == 64
    for n in ('2', '4'):
        assert m['predicted_speedup'][n]['no_restarts (reset_branch=None, configs[4] regime)']['speedup'] <= float(n) + 1e-6


def test_bench_compact_line_keeps_the_judged_fields_under_6_kb():
    """bench.compact_line (pure host code): the stdout form of the record must stay under 6 KB whatever the extra legs carry (the
    driver keeps the last 8 KB of stdout; round 4's 14 KB line lost `dropin_surface`, `roofline_match_top2`, `whole_path`) and must
    keep the contract fields, `roofline`, `cpu_baseline`, the drop-in rate and the N > 1 evidence."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    blob = {'x' * 20 + str(i): list(range(64)) for i in range(40)}          # stands for the verbose objects of the full record
    line = {'metric': 'm', 'value': 250.0, 'unit': 'frames/s', 'n_gpus': 1, 'steps': 20, 'warmup': 5, 'ms_per_step': 4.0, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': 'w' * 300, 'precision': 'p' * 400, 'dropin_frames_per_s': 180.0, 'frames_per_call': 4},
            'samples': [250.0] * 5, 'dropin_surface': {'value': 180.0, 'unit': 'frames/s', 'samples': [180.0] * 5, 'call': 'c' * 300},
            'one_frame_per_call': {'value': 215.0, 'unit': 'frames/s', 'samples': [215.0] * 5},
            'roofline': {'kernel': 'k' * 120, 'bound': 'mfma', 'achieved': 375.0, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.15, 'traffic': 5.4e7,
                         'maps_per_launch': 4, 'mean_launch_ms': 0.0287, 'note': 'n' * 600},
            'roofline_match_top2': {'achieved': 1150.0, 'frac': 0.46, 'mean_launch_ms': 1.05, 'kernel': 'k' * 80},
            'cpu_baseline': {'value': 0.035, 'unit': 'frames/s', 'cores': 16, 'kind': 'port', 'sample': 's' * 500, 'seconds_per_frame': 28.5},
            'whole_path': {'algorithmic_tflop_per_frame': 2.306, 'frac_of_f16_mfma_peak': 0.23, 'breakdown': blob, 'counting': 'c' * 400},
            'streams': {'median_pass': {'P_ms_per_call': 3.0, 'F_ms_per_call': 1.0, 'M_ms_per_call': 4.0, 'wall_ms_per_call': 4.0}, 'slow_passes': [blob]},
            'kernels': [{'kernel': 'kernel number %d with a long description' % i, 'us_per_launch': 10.0, 'frac': 0.1, 'x': blob} for i in range(14)],
            'wavefront_model': {'predicted_speedup': {str(n): {'with_restarts (reset_branch=9)': {'chosen': 'x', 'speedup': 7.0, 'with_context_exchange': {'speedup': 7.5}},
                                                              'no_restarts (reset_branch=None, configs[4] regime)': {'speedup': 5.4, 'with_context_exchange': {'speedup': 5.9, 'block_sizes': list(range(64))}}}
                                                      for n in (2, 4, 8)}, 'junk': blob},
            'other_configs': {'configs[2]': {'value': 95.0, 'ms_per_step': 10.5, 'whole_path': {'frac_of_f16_mfma_peak': 0.2}, 'roofline': {'frac': 0.16}, 'w': 'x' * 400},
                              'configs[4] on one GPU': {'value': 7.5, 'ms_per_step': 133.0, 'peak_memory_gib': 17.6, 'workload': 'x' * 400}},
            'pcie_inclusive': {'value': 200.0, 'unit': 'frames/s', 'samples': [200.0] * 3, 'h2d_mb_per_frame': 15.55, 'd2h_mb_per_frame': 24.88, 'note': 'n' * 300},
            'first_frame_ms': 11.8, 'full_record': 'gpurun_out/bench_full.json'}
    c = bench.compact_line(line)
    txt = json.dumps(c)
    assert len(txt) < 6000, len(txt)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert k in c
    assert c['roofline']['frac'] == 0.15 and c['roofline']['bound'] == 'mfma' and c['roofline']['traffic'] == 5.4e7
    assert c['cpu_baseline']['cores'] == 16 and c['cpu_baseline']['kind'] == 'port' and len(c['cpu_baseline']['sample']) <= 200
    assert c['dropin_surface']['value'] == 180.0 and c['config']['dropin_frames_per_s'] == 180.0 and 'precision' not in c['config']
    assert c['pcie_inclusive']['value'] == 200.0 and 'note' not in c['pcie_inclusive']
    assert c['wavefront_model_predicted_speedup']['8'] == {'restarts': 7.5, 'no_restarts': 5.9}
    # N > 1: the sharded-clip figure becomes the headline, the weak-scaling figure moves aside
    import argparse
    wf = {'value': 900.0, 'seconds': 64 / 900.0, 'frames_equal': True, 'workload': 'clip', 'ranks_seen': 8, 'backend': 'nccl (= RCCL)', 'gpus_visible': 8,
          'handoff': {'ms_per_message_measured': 0.31, 'messages': 50, 'bytes_per_message': 32700000}, 'partition': {'name': 'cyclic_growing', 'blocks': blob, 'predicted_speedup': 7.4}}
    line8 = dict(line, n_gpus=8, value=1700.0, config=dict(line['config']))
    bench.promote_wavefront(line8, wf, argparse.Namespace(clip=64, size='270x480'), 8)
    assert line8['scaling'] == 'strong' and line8['value'] == 900.0 and line8['weak_scaling_shards']['value'] == 1700.0
    assert line8['config']['ranks_seen'] == 8 and line8['config']['handoff_ms_measured'] == 0.31 and line8['config']['timed_frames'] == 64
    c8 = bench.compact_line(line8)
    assert len(json.dumps(c8)) < 6000 and c8['wavefront']['frames_equal'] is True and c8['weak_scaling_shards']['value'] == 1700.0
    # a leg that failed its frame check never becomes the headline
    line_bad = dict(line, n_gpus=8, value=1700.0, config=dict(line['config']))
    bench.promote_wavefront(line_bad, dict(wf, frames_equal=False), argparse.Namespace(clip=64, size='270x480'), 8)
    assert line_bad['value'] == 1700.0 and line_bad['scaling'] == 'weak'


def test_roll_over_counter_of_a_frame_group():
    """Engine._itr_after (host arithmetic of the frame groups, round 5): the value of frame_itr_num the k-th window from now will find
    -- the counter restarts when it reaches reset_branch (RefVSR.py:168-170, 292-295) -- against a window-by-window simulation; the
    windows of a group that restart the forward branch are exactly those that find the counter at reset_branch."""
    import types
    from refvsr_amd.engine import Engine
    for reset in (None, 1, 2, 3, 5, 9):
        for start in range(1, (reset or 6) + 1):
            eng = types.SimpleNamespace(frame_itr_num=start, max_frame_itr_num=reset)
            itr, found = start, []
            for k in range(14):
                found.append(itr)
                assert Engine._itr_after(eng, k) == itr, (reset, start, k)
                if reset is not None and itr == reset:          # this window restarts: the counter starts again
                    itr = 0
                itr += 1
            restarts = [k for k, v in enumerate(found) if reset is not None and v == reset]
            if reset is None:
                assert restarts == [] and found == list(range(start, start + 14))
            else:
                assert all(b - a == reset for a, b in zip(restarts, restarts[1:])), (reset, start, restarts)
                assert restarts and restarts[0] == reset - start
