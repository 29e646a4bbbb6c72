"""The oracle (oracle/refvsr_oracle.py) against fixtures produced by the REAL reference
(tools/gen_golden.py, run in the build container).  This is what pins the oracle."""
import pytest
import torch

from conftest import load_golden, maxdiff
from oracle import refvsr_oracle as orc
from refvsr_amd import get_config, make_state_dict
from refvsr_amd.synth import window_indices

TOL = 2e-5      # fp32 restatement vs reference; measured deltas are <= 1.1e-5 (see gen_golden output)


def test_warp_and_flow_warp():
    g = load_golden('op_warp')
    assert maxdiff(orc.warp(g['x'], g['flow']), g['warp']) < TOL
    assert maxdiff(orc.warp(g['x'], g['flow2']), g['warp2']) < TOL      # LR input, 2x flow (RefVSR.py:254)
    assert maxdiff(orc.flow_warp_border(g['x'], g['flow']), g['flow_warp']) < TOL


def test_resize_modes():
    g = load_golden('op_resize')
    img, fl = g['img'], g['flow']
    assert maxdiff(orc.bicubic_scale(img, 0.5, False), g['bicubic_half']) < TOL
    assert maxdiff(orc.bicubic_scale(img, 2, False), g['bicubic_x2']) < TOL
    assert maxdiff(orc.bicubic_scale(img, 4, False), g['bicubic_x4']) < TOL
    assert maxdiff(orc.flow_up2(fl), g['flow_up2']) < TOL
    assert maxdiff(orc.resize(img, (32, 32), 'bilinear'), g['bilinear_32x32']) < TOL
    assert maxdiff(orc.resize(g['bilinear_32x32'], (18, 26), 'bilinear'), g['bilinear_back']) < TOL
    assert maxdiff(orc.resize(img, (9, 13), 'nearest', 2.0), g['nearest_half']) == 0.0


def test_patches_reflect():
    g = load_golden('op_patches')
    assert maxdiff(orc.patches3x3(g['f']), g['patches']) == 0.0


def test_feature_match(small_sd):
    g = load_golden('op_match')
    conf, idx = orc.feature_match(g['lr'], g['ref'], small_sd, False)
    assert maxdiff(conf, g['conf']) < TOL
    assert torch.equal(idx, g['idx'])
    # chunked evaluation gives identical per-column results
    conf2, idx2 = orc.feature_match(g['lr'], g['ref'], small_sd, False, chunk=100)
    assert maxdiff(conf2, g['conf']) < TOL and torch.equal(idx2, g['idx'])


def test_argmax_first_index_on_ties():
    ref_p = torch.zeros(1, 6, 4)
    ref_p[0, 2, 0] = ref_p[0, 4, 0] = 1.0
    lr_p = torch.zeros(1, 4, 3)
    lr_p[0, 0, :] = 1.0
    v, i = orc.match_argmax(ref_p, lr_p)
    assert i.tolist() == [[2, 2, 2]]


def test_block_gather_and_aligned_conv(small_sd):
    g = load_golden('op_aa')
    idx = g['idx']
    assert maxdiff(orc.block_gather(g['value_down'], idx, 1, (20, 28)), g['aa1']) == 0.0
    assert maxdiff(orc.block_gather(g['value'], idx, 2, (40, 56)), g['aa2_fm']) == 0.0
    rgb = orc.block_gather(g['ref'], idx, 2, (40, 56))
    assert maxdiff(rgb, g['aa2_rgb']) == 0.0
    out = orc.aligned_conv(g['aa2_fm'], g['lr'], rgb, small_sd, 'Network.aa2.align', 2)
    assert maxdiff(out, g['aa2']) < TOL


def test_aligned_sampler():
    g = load_golden('op_sampler')
    assert maxdiff(orc.aligned_sample(g['x'], g['affine'], 2), g['out']) < TOL
    # zero predictor output (affine == 1) is the identity
    x = torch.rand(1, 2, 8, 12)
    for ks in (2, 4):
        aff = torch.ones(1, 3, 8 // ks, 12 // ks)
        assert maxdiff(orc.aligned_sample(x, aff, ks), x) < 1e-6


def test_spynet(small_sd):
    g = load_golden('op_spynet')
    assert maxdiff(orc.spynet(g['a'], g['b'], small_sd), g['flow']) < TOL


def test_conv_stacks(small_cfg, small_sd):
    g = load_golden('op_convs')
    assert maxdiff(orc.res_list(g['feat'], small_sd, 'Network.feat_decoder2', 4), g['res_list']) < TOL
    x = torch.cat([g['img'], g['feat']], 1)
    assert maxdiff(orc.resblocks_with_input_conv(x, small_sd, 'Network.backward_resblocks', small_cfg.num_blocks),
                   g['resblocks']) < TOL
    assert maxdiff(orc.pixel_shuffle_pack(g['feat'], small_sd, 'Network.upsample1'), g['pixel_shuffle']) < TOL


def test_compute_up(small_cfg, small_sd):
    g = load_golden('op_compute_up')
    o = orc.OracleNetwork(small_cfg, small_sd)
    assert maxdiff(o._compute_up(g['bw'], g['fw'], g['conf_bw'], g['conf_fw'], g['base']), g['out']) < TOL


E2E = [('S_16x16_t3', 'config_RefVSR_small_L1'), ('S_18x26_t5', 'config_RefVSR_small_L1'),
       ('S_24x32_t5_reset3', 'config_RefVSR_small_L1'), ('F_16x24_t3', 'config_RefVSR_MFID'),
       ('HD_32x48_t3', 'config_RefVSR_small_MFID_8K'), ('S_16x24_t7', 'config_RefVSR_small_L1'),
       ('HD48_64x96_t3', 'config_RefVSR_MFID_8K'), ('S2_16x24_t3', 'config_RefVSR_small_L1')]


@pytest.mark.parametrize('tag,name', E2E)
def test_end_to_end_stream(tag, name):
    """First-frame call, chained steady-state calls and a reset_branch rollover: result, the four
    carried state tensors, the iteration counter and the eval_vis confidence maps."""
    g = load_golden('e2e_' + tag)
    cfg = get_config('p', 'm', name)
    t = int(g['t'])
    cfg.frame_num = t
    cfg.save_sample = True
    rb = int(g['reset_branch'])
    cfg.reset_branch = None if rb < 0 else rb
    if int(g.get('scale', 4)) != 4:                      # x2 fixtures: the reference ran with `config.scale = 2`
        from refvsr_amd import set_scale
        set_scale(cfg, int(g['scale']))
    o = orc.OracleNetwork(cfg, make_state_dict(cfg, 1234))
    lr, rf = g['lr'], g['ref']
    nframes = lr.shape[1]
    for f in range(nframes):
        w = window_indices(f, nframes, t)
        outs = o.forward(lr[:, w], rf[:, w], f == 0, is_log=True)
        assert maxdiff(outs['result'], g['result_%d' % f]) < TOL, (tag, f)
        assert o.frame_itr_num == int(g['itr_%d' % f])
        f_tol = TOL if g['state_feat_%d' % f].dtype == torch.float32 else 2e-3       # big maps stored as fp16
        assert maxdiff(o.forward_feat_prop_prev, g['state_feat_%d' % f]) < f_tol
        if ('state_feat_up_%d' % f) in g:                                            # (left out of the light fixtures)
            up_tol = TOL if g['state_feat_up_%d' % f].dtype == torch.float32 else 2e-3
            assert maxdiff(o.forward_feat_prop_UP_prev, g['state_feat_up_%d' % f]) < up_tol
        assert maxdiff(o.forward_conf_map_prop_prev, g['state_conf_%d' % f]) < TOL
        assert maxdiff(o.forward_flow_prev, g['state_flow_%d' % f]) < TOL
        for k, v in outs['eval_vis'].items():
            assert maxdiff(v, g['ev_%s_%d' % (k, f)]) < TOL, (tag, f, k)


def test_refvsr_ir_stream():
    """RefVSR_IR oracle (oracle/refvsr_ir_oracle.py: EDVR-M extractor, PCD alignment, modulated deformable conv, TSA
    fusion, key-frame refill, the forward branch's stale-flow quirk) against the fixture produced by the reference."""
    from oracle import refvsr_ir_oracle as iro
    g = load_golden('e2e_IR_64x64_t5_reset2')
    cfg = get_config('p', 'm', 'config_RefVSR_IR_MFID')
    t = int(g['t'])
    cfg.frame_num = t
    cfg.reset_branch = int(g['reset_branch'])
    cfg.save_sample = True
    o = iro.OracleNetworkIR(cfg, make_state_dict(cfg, 1234))
    lr, rf = g['lr'], g['ref']
    for f in range(2):                      # first-frame + steady call (the other two repeat them after the reset)
        w = window_indices(f, lr.shape[1], t)
        outs = o.forward(lr[:, w], rf[:, w], f == 0, is_log=True)
        assert maxdiff(outs['result'], g['result_%d' % f]) < TOL
        assert o.frame_itr_num == int(g['itr_%d' % f]) and list(o.keyframe_idx) == g['keyframes_%d' % f].tolist()
        # the `vis` samples (RefVSR_IR.py:367-384): same keys in the same order, values
        vkeys = [k[4:-2] for k in g if k.startswith('vis_') and k.endswith('_%d' % f)]
        assert sorted(outs['vis'].keys()) == sorted(vkeys) and len(vkeys) == 7
        for k in vkeys:
            assert maxdiff(outs['vis'][k], g['vis_%s_%d' % (k, f)]) < 1e-4, (f, k)
        assert maxdiff(o.forward_feat_prop_prev, g['state_feat_%d' % f].float()) < 2e-3      # stored as fp16
        assert maxdiff(o.forward_flow_prev, g['state_flow_%d' % f]) < TOL


def test_ir_oracle_deformable_conv_against_independent_formulations():
    """The modulated deformable convolution inside the IR fixture is the oracle's own restatement of mmcv's compiled op (absent from this
    image: the op stays "parity-unpinned" against mmcv itself).  What CAN be pinned here: (1) its bilinear sampler against torch's
    F.grid_sample(bilinear, zeros, align_corners=True) -- the same definition as mmcv's dmcn_im2col_bilinear: a sample is 0 outside
    (-1, H) x (-1, W), corner pixels outside the map contribute 0 -- on coordinates that leave the map on every side; (2) the whole
    ModulatedDCNPack against plain convolutions where the op degenerates to one: zero offsets (0.5 x conv: sigmoid(0) masks) and
    integer offsets (the conv of the shifted, zero-filled map)."""
    import torch.nn.functional as F
    from oracle import refvsr_ir_oracle as iro
    g = torch.Generator().manual_seed(7)
    n, c, H, W = 2, 8, 13, 17
    x = torch.randn(n, c, H, W, generator=g, dtype=torch.float64)
    py = torch.rand(n, 1, H, W, generator=g, dtype=torch.float64) * (H + 6) - 3.0          # [-3, H + 3): leaves the map on every side
    px = torch.rand(n, 1, H, W, generator=g, dtype=torch.float64) * (W + 6) - 3.0
    py[0, 0, 0, :4] = torch.tensor([-1.0, 0.0, H - 1.0, float(H)], dtype=torch.float64)    # the boundaries themselves
    px[0, 0, 0, :4] = torch.tensor([-1.0, W - 1.0, float(W), 0.0], dtype=torch.float64)
    got = iro.deform_sample(x, py, px)
    grid = torch.stack([2.0 * px[:, 0] / (W - 1) - 1.0, 2.0 * py[:, 0] / (H - 1) - 1.0], -1)
    want = F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=True)
    assert maxdiff(got, want) < 1e-12
    # the pack: zero conv_offset -> every tap at its integer position with mask 0.5
    M = 16
    xs = torch.randn(1, M, 9, 11, generator=g)
    extra = torch.randn(1, M, 9, 11, generator=g)
    Wt = {'d.weight': torch.randn(M, M, 3, 3, generator=g) / 12.0, 'd.bias': torch.randn(M, generator=g) * 0.1,
          'd.conv_offset.weight': torch.zeros(216, M, 3, 3), 'd.conv_offset.bias': torch.zeros(216)}
    want0 = 0.5 * F.conv2d(xs, Wt['d.weight'], None, padding=1) + Wt['d.bias'].view(1, -1, 1, 1)
    assert maxdiff(iro.dcn_pack(xs, extra, Wt, 'd'), want0) < 1e-5
    # integer offsets (dy, dx) for every tap and group, mask logits 0: 0.5 x the conv of the map shifted by (dy, dx), zero filled
    dy, dx = 2, -3
    bias = torch.zeros(216)
    bias[0:144:2] = float(dy)               # channel g*18 + 2k: row offset, + 1: column offset
    bias[1:144:2] = float(dx)
    Wt['d.conv_offset.bias'] = bias
    sh = torch.zeros_like(xs)
    sh[:, :, max(0, -dy):xs.shape[2] - max(0, dy), max(0, -dx):xs.shape[3] - max(0, dx)] = \
        xs[:, :, max(0, dy):xs.shape[2] - max(0, -dy), max(0, dx):xs.shape[3] - max(0, -dx)]
    # tap k of output pixel (y, x) reads x[y + ky - 1 + dy, x + kx - 1 + dx] (0 outside the map) = the zero-padded conv of the shifted map,
    # except where the SHIFTED map's zero fill and the conv's zero padding disagree with "0 outside the original map": they agree everywhere
    want1 = 0.5 * F.conv2d(sh, Wt['d.weight'], None, padding=1) + Wt['d.bias'].view(1, -1, 1, 1)
    got1 = iro.dcn_pack(xs, extra, Wt, 'd')
    # rows / columns whose 3x3 window reaches shifted-in data beyond the padding ring differ by construction; compare the interior
    assert maxdiff(got1[:, :, 3:-3, 4:-4], want1[:, :, 3:-3, 4:-4]) < 1e-5
