"""Algorithmic work of the RefVSR inference path (multiply-add = 2 FLOP), counted from the layer list of
/root/reference/models/archs/RefVSR.py -- the figure `bench.py` divides by the measured time.

`dedup=True` (the roofline figure): work that the reference recomputes identically across overlapping
sliding windows -- matching, reference encoders and alignment of a frame already seen, flows already computed -- is
counted once per output frame: 2 SPyNet calls, 1 matching, 1 encoder / alignment pass, 3 backward + 1 forward
propagation steps (with their RAP fusion), 1 upsampler.  `dedup=False`: what one steady-state call of the reference
executes (2(t-1) SPyNet calls, t//2+1 matchings, encoders / alignment inside every propagation step).
Validated against SURVEY.md 8d (FlopCounterMode on the reference): S @270x480 t=5: 2.490 (dedup) / 5.517 (as executed)."""


def _conv(cin, cout, k, px):
    return 2.0 * cin * cout * k * k * px


def spynet_call(h, w):
    h_up = h if h % 32 == 0 else 32 * (h // 32 + 1)
    w_up = w if w % 32 == 0 else 32 * (w // 32 + 1)
    per_px = sum(_conv(ci, co, 7, 1) for ci, co in ((8, 32), (32, 64), (64, 32), (32, 16), (16, 2)))
    return per_px * sum((h_up >> l) * (w_up >> l) for l in range(6))


def matching(cfg, h, w):
    """(GEMM, VGG head) of one FeatureMatching.forward."""
    if cfg.flag_HD_in:
        f = cfg.scale // 2
        hl, wl = h // f, w // f                       # nearest x(1/f), then VGG19[0:7] with a 2x max-pool
        vgg = lambda px: _conv(3, 64, 3, px) + _conv(64, 64, 3, px) + _conv(64, 128, 3, px // 4) + _conv(128, 16, 1, px // 4)
        n_lr, n_ref = (hl // 2) * (wl // 2), (hl // 4) * (wl // 4)
        return 2.0 * n_lr * n_ref * 144, vgg(hl * wl) + vgg((hl // 2) * (wl // 2))
    if cfg.scale != 4:                                # x2: VGG19[0:7] features, matching grid = half the frame
        vgg = lambda px: _conv(3, 64, 3, px) + _conv(64, 64, 3, px) + _conv(64, 128, 3, px // 4) + _conv(128, 16, 1, px // 4)
        n_lr, n_ref = (h // 2) * (w // 2), (h // 4) * (w // 4)
        return 2.0 * n_lr * n_ref * 144, vgg(h * w) + vgg((h // 2) * (w // 2))
    vgg = lambda px: _conv(3, 64, 3, px) + _conv(64, 64, 3, px) + _conv(64, 16, 1, px)
    n_lr, n_ref = h * w, (h // 2) * (w // 2)
    return 2.0 * n_lr * n_ref * 144, vgg(n_lr) + vgg(n_ref)


def aligned_conv(px_in, ks):
    """AlignedConv2d predictor on a px_in-pixel map sampled with stride ks (alignment.py:18-24)."""
    enc = _conv(3, 32, 5, px_in) + 2 * _conv(32, 32, 3, px_in)
    px_o = px_in // (ks * ks)
    return 2 * enc + _conv(64, 32, 5, px_o) + 2 * _conv(32, 32, 3, px_o) + _conv(32, 3, 1, px_o)


def per_frame_prepare(cfg, h, w):
    """Reference encoders + both AlignedAttention predictors of one frame (functions of (lr_i, ref_i) only)."""
    C, LR = cfg.mid_channels, h * w
    enc = _conv(3, C, 3, LR) + _conv(C, C, 3, LR) + 9 * _conv(C, C, 3, LR)                       # ref_encoder1 + res1
    enc += _conv(C, C, 3, LR // 4) + _conv(C, C, 3, LR // 4) + 9 * _conv(C, C, 3, LR // 4)       # ref_encoder2 (stride 2) + res2
    s1, s2 = cfg.matching_ksize // 2, cfg.matching_ksize
    al = aligned_conv(4 * LR, s2)                                                                # aa2 on the 2x map
    if s1 > 1:
        al += aligned_conv(LR, s1)                                                               # aa1 (HD configs)
    return enc + al


def propagation_step(cfg, h, w):
    """ResidualBlocksWithInputConv + the convs of AA_AF_conf_prop (RefVSR.py:123-149,327-360) without the per-frame part."""
    C, nb, LR, X2 = cfg.mid_channels, cfg.num_blocks, h * w, 4 * h * w
    f = _conv(C + 3, C, 3, LR) + 2 * nb * _conv(C, C, 3, LR)
    f += _conv(2, 16, 3, LR) + _conv(16, C, 3, LR) + _conv(2 * C, C, 3, LR) + _conv(C, C, 3, LR) + 17 * _conv(C, C, 3, LR)
    f += _conv(C, 4 * C, 3, LR) + _conv(2 * C, C, 3, X2)
    f += _conv(2, 16, 3, X2) + _conv(16, C, 3, X2) + _conv(2 * C, C, 3, X2) + _conv(C, C, 3, X2) + 9 * _conv(C, C, 3, X2)
    return f


def upsampler(cfg, h, w):
    C, X2 = cfg.mid_channels, 4 * h * w
    HR = 16 * h * w if cfg.scale == 4 else X2
    f = _conv(2 * C, C, 1, X2) + _conv(2, 16, 3, X2) + _conv(16, C, 3, X2) + _conv(2 * C, C, 3, X2) + _conv(C, C, 3, X2)
    f += 9 * _conv(C, C, 3, X2) + (_conv(C, 4 * C, 3, X2) if cfg.scale == 4 else 0.0) + _conv(C, C, 3, HR) + _conv(C, 3, 3, HR)
    return f


def tflop_per_frame(cfg, h, w, t=5, dedup=True):
    """Algorithmic TFLOP per steady-state output frame; returns (total, breakdown dict)."""
    gemm, vgg = matching(cfg, h, w)
    steps = (t - 1 - t // 2 + 1) + 1                       # backward steps over frames t-1..ctr, one forward step
    if dedup:
        parts = dict(match_gemm=gemm, match_vgg=vgg, spynet=2 * spynet_call(h, w), prepare=per_frame_prepare(cfg, h, w),
                     propagation=steps * propagation_step(cfg, h, w), upsampler=upsampler(cfg, h, w))
    else:
        nm = t // 2 + 1
        parts = dict(match_gemm=nm * gemm, match_vgg=nm * vgg, spynet=2 * (t - 1) * spynet_call(h, w),
                     prepare=steps * per_frame_prepare(cfg, h, w), propagation=steps * propagation_step(cfg, h, w),
                     upsampler=upsampler(cfg, h, w))
    parts = {k: v / 1e12 for k, v in parts.items()}
    return sum(parts.values()), parts
