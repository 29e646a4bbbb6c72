"""CPU ORACLE for the RefVSR inference hot path -- TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  `refvsr_amd/` never does.

It is an independent fp32 restatement (plain torch CPU ops + explicit index math) of
`Network.forward` of the reference and everything below it.  Every function cites the reference
file:line it follows (paths relative to /root/reference).  Resampling ops (warp, flow_warp,
bicubic / bilinear / nearest resize, block gather, affine patch sampler) are written out from
their maths -- they do NOT call F.grid_sample / F.interpolate / unfold / fold -- so that the HIP
kernels, which are written from the same maths, are checked against something that was itself
pinned against the real reference.

Pinning: the reference has no tests or golden vectors.  The oracle is pinned
against outputs of the reference itself, imported in the build container by
tools/gen_golden.py (fixtures in tests/golden/*.npz, checked by tests/test_oracle_golden.py).
Third-party boundaries of the reference (mmcv.ConvModule == Conv2d+ReLU, torchvision VGG19 layer
list, ATen resampling semantics of torch 2.10 vs the 1.8-1.11 the reference pins) are NOT
covered by any reference test: parity there is unpinned.
"""
import collections
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------------------------


def conv(x, W, name, stride=1, pad=None):
    """nn.Conv2d with bias, zero padding k//2 (common.py:7-10; nn.Conv2d call sites)."""
    w = W[name + '.weight']
    b = W[name + '.bias']
    if pad is None:
        pad = w.shape[-1] // 2
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def lrelu(x, slope):
    return torch.where(x >= 0, x, x * slope)


def _cubic_w(t, A=-0.75):
    """Cubic-convolution taps for fractional offset t (ATen upsample_bicubic2d, A=-0.75)."""
    def near(x):   # |x| <= 1
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

    def far(x):    # 1 < |x| < 2
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A
    return far(t + 1.0), near(t), near(1.0 - t), far(2.0 - t)


def resize_matrix(n_in, n_out, mode, src_scale=None):
    """Dense [n_out, n_in] 1-D resampling operator with PyTorch F.interpolate semantics.

    mode: 'bicubic' (align_corners=False, A=-0.75, border-replicated taps),
          'bilinear' (align_corners=False), 'bilinear_ac' (align_corners=True), 'nearest'.
    src_scale: source step per output sample; 1/scale_factor when interpolate() was called with
          scale_factor (RefVSR.py:105,125,140,288; alignment.py:41), n_in/n_out when called with
          size= (SPyNet.py:120-133).
    """
    M = torch.zeros(n_out, n_in, dtype=torch.float64)
    s = float(n_in) / float(n_out) if src_scale is None else float(src_scale)
    for i in range(n_out):
        if mode == 'bicubic':
            x = (i + 0.5) * s - 0.5
            ix = math.floor(x)
            t = x - ix
            for k, wk in enumerate(_cubic_w(t)):
                j = min(max(ix - 1 + k, 0), n_in - 1)
                M[i, j] += wk
        elif mode == 'bilinear':
            x = max((i + 0.5) * s - 0.5, 0.0)
            i0 = min(int(math.floor(x)), n_in - 1)
            i1 = min(i0 + 1, n_in - 1)
            l1 = x - i0
            M[i, i0] += 1.0 - l1
            M[i, i1] += l1
        elif mode == 'bilinear_ac':
            sc = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
            x = i * sc
            i0 = min(int(math.floor(x)), n_in - 1)
            i1 = min(i0 + 1, n_in - 1)
            l1 = x - i0
            M[i, i0] += 1.0 - l1
            M[i, i1] += l1
        elif mode == 'nearest':
            j = min(int(math.floor(i * s)), n_in - 1)
            M[i, j] = 1.0
        else:
            raise ValueError(mode)
    return M.to(torch.float32)


_RM_CACHE = {}


def _rm(n_in, n_out, mode, src_scale):
    key = (n_in, n_out, mode, src_scale)
    if key not in _RM_CACHE:
        _RM_CACHE[key] = resize_matrix(n_in, n_out, mode, src_scale)
    return _RM_CACHE[key]


def resize(x, out_hw, mode, src_scale=None):
    """Separable resize of [n,c,h,w] (rows first, then columns: same 2-D weights as ATen)."""
    h, w = x.shape[-2:]
    Mh = _rm(h, out_hw[0], mode, src_scale)
    Mw = _rm(w, out_hw[1], mode, src_scale)
    return torch.matmul(torch.matmul(Mh, x), Mw.t())


def bicubic_scale(x, factor, clamp=True):
    """F.interpolate(x, scale_factor=factor, mode='bicubic', align_corners=False)[.clamp(0,1)]."""
    h, w = x.shape[-2:]
    oh, ow = int(math.floor(h * factor)), int(math.floor(w * factor))
    y = resize(x, (oh, ow), 'bicubic', 1.0 / factor)
    return y.clamp(0, 1) if clamp else y


def flow_up2(flow):
    """F.interpolate(flow, scale_factor=2, bilinear, align_corners=True) * 2 (RefVSR.py:220,254,259;
    SPyNet.py:88-92)."""
    h, w = flow.shape[-2:]
    return resize(flow, (2 * h, 2 * w), 'bilinear_ac') * 2.0


def avg_pool2(x):
    """2x2/2 average pooling (SPyNet.py:68-79; attention.py:51)."""
    h, w = x.shape[-2:]
    x = x[..., :h // 2 * 2, :w // 2 * 2]
    return 0.25 * (x[..., 0::2, 0::2] + x[..., 0::2, 1::2] + x[..., 1::2, 0::2] + x[..., 1::2, 1::2])


def max_pool2(x):
    h, w = x.shape[-2:]
    x = x[..., :h // 2 * 2, :w // 2 * 2]
    return torch.maximum(torch.maximum(x[..., 0::2, 0::2], x[..., 0::2, 1::2]),
                         torch.maximum(x[..., 1::2, 0::2], x[..., 1::2, 1::2]))


def _bilinear_gather(x, xs, ys, zero_pad):
    """Sample x[n,c,H,W] at float pixel coords xs,ys [n,Ho,Wo]; out-of-range taps are 0 when
    zero_pad, else the caller has already clamped the coordinates."""
    n, c, H, W = x.shape
    x0 = torch.floor(xs)
    y0 = torch.floor(ys)
    tx = xs - x0
    ty = ys - y0
    x0 = x0.long()
    y0 = y0.long()
    flat = x.reshape(n, c, H * W)
    out = torch.zeros(n, c, xs.shape[1], xs.shape[2], dtype=x.dtype)
    for dy, wy in ((0, 1.0 - ty), (1, ty)):
        for dx, wx in ((0, 1.0 - tx), (1, tx)):
            xi = x0 + dx
            yi = y0 + dy
            valid = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            xi = xi.clamp(0, W - 1)
            yi = yi.clamp(0, H - 1)
            idx = (yi * W + xi).reshape(n, 1, -1).expand(-1, c, -1)
            v = torch.gather(flat, 2, idx).reshape(n, c, xs.shape[1], xs.shape[2])
            wgt = wy * wx
            if zero_pad:
                wgt = wgt * valid.to(x.dtype)
            out = out + v * wgt[:, None]
    return out


def warp(x, flow):
    """Inter-frame alignment (models/utils.py:35-43).

    Base grid linspace(-1,1,Wf) (an align_corners=True style grid) + flow/((Win-1)/2), sampled with
    grid_sample(bilinear, zeros, align_corners=False):  x_src = ((g+1)*Win - 1)/2.
    Output size = flow size (the 2x-flow / LR-input form of RefVSR.py:254 included)."""
    n, c, Hin, Win = x.shape
    Hf, Wf = flow.shape[-2:]
    gx = torch.linspace(-1.0, 1.0, Wf).view(1, 1, Wf) + flow[:, 0] / ((Win - 1.0) / 2.0)
    gy = torch.linspace(-1.0, 1.0, Hf).view(1, Hf, 1) + flow[:, 1] / ((Hin - 1.0) / 2.0)
    xs = ((gx + 1.0) * Win - 1.0) / 2.0
    ys = ((gy + 1.0) * Hin - 1.0) / 2.0
    return _bilinear_gather(x, xs, ys, zero_pad=True)


def flow_warp_border(x, flow):
    """mmedit flow_warp(x, flow, 'bilinear', padding_mode='border', align_corners=True)
    (mmedit/models/common/flow_warp.py:6-47): x_src = j + u, clamped to the
    border.  `flow` is [n,2,h,w] (ch0 = x / u)."""
    n, c, H, W = x.shape
    jj = torch.arange(W, dtype=x.dtype).view(1, 1, W)
    ii = torch.arange(H, dtype=x.dtype).view(1, H, 1)
    # normalise / unnormalise exactly as the reference does (keeps fp32 rounding comparable)
    gx = 2.0 * (jj + flow[:, 0]) / max(W - 1, 1) - 1.0
    gy = 2.0 * (ii + flow[:, 1]) / max(H - 1, 1) - 1.0
    xs = ((gx + 1.0) / 2.0 * (W - 1)).clamp(0, W - 1)
    ys = ((gy + 1.0) / 2.0 * (H - 1)).clamp(0, H - 1)
    return _bilinear_gather(x, xs, ys, zero_pad=False)


def reflect_index(i, n):
    """ReflectionPad2d index map (no edge repeat): -1 -> 1, n -> n-2."""
    i = torch.where(i < 0, -i, i)
    return torch.where(i >= n, 2 * (n - 1) - i, i)


def patches3x3(f):
    """extract_image_patches(f, 3x3, stride 1, 'same') (RefVSR_/utils.py:10-57): reflection pad 1
    then unfold; channel order c*9 + ky*3 + kx.  Returns [n, c*9, H*W]."""
    n, c, H, W = f.shape
    ys = reflect_index(torch.arange(-1, H + 1), H)
    xs = reflect_index(torch.arange(-1, W + 1), W)
    fp = f[:, :, ys][:, :, :, xs]                       # [n,c,H+2,W+2]
    cols = []
    for ky in range(3):
        for kx in range(3):
            cols.append(fp[:, :, ky:ky + H, kx:kx + W])
    p = torch.stack(cols, 2)                              # [n,c,9,H,W]
    return p.reshape(n, c * 9, H * W)


def block_gather(value, index_map, s, out_hw):
    """AlignedAttention without `align` (attention.py:142-144,118-128).

    unfold(k=s, stride=s) -> gather along L by index_map -> fold(k=s, stride=s) to out_hw is a
    pure block gather: out[c, s*y+ky, s*x+kx] = value[c, s*ry+ky, s*rx+kx],
    (ry,rx) = divmod(idx[y*w+x], Wv//s) with w = out_w//s.  (Wv//s is the patch-grid width of
    *this* value tensor -- it intentionally reproduces the HD aa1 RGB quirk where it differs from
    the index grid.)"""
    n, c, Hv, Wv = value.shape
    oh, ow = out_hw
    gh, gw = oh // s, ow // s
    Wr = Wv // s
    idx = index_map.reshape(n, gh, gw)
    ry = idx // Wr
    rx = idx % Wr
    ky = torch.arange(s).view(1, 1, s, 1, 1)
    kx = torch.arange(s).view(1, 1, 1, 1, s)
    sy = (ry[:, :, None, :, None] * s + ky).expand(n, gh, s, gw, s).reshape(n, oh * ow)
    sx = (rx[:, :, None, :, None] * s + kx).expand(n, gh, s, gw, s).reshape(n, oh * ow)
    lin = (sy * Wv + sx)[:, None, :].expand(-1, c, -1)
    return torch.gather(value.reshape(n, c, Hv * Wv), 2, lin).reshape(n, c, oh, ow)


# --------------------------------------------------------------------------------------------
# network blocks
# --------------------------------------------------------------------------------------------

def res_block(x, W, name):
    """RefVSR_/common.py:25-39: x + conv2(LeakyReLU0.2(conv1(x)))."""
    return x + conv(lrelu(conv(x, W, name + '.conv1'), 0.2), W, name + '.conv2')


def res_list(x, W, name, n):
    """RefVSR_/common.py:64-82."""
    y = x
    for i in range(n):
        y = res_block(y, W, '%s.RBs.%d' % (name, i))
    return conv(y, W, name + '.conv_tail') + x


def basic2(x, W, name, stride0=1):
    """Two stacked BasicBlocks conv+LeakyReLU(0.2) (RefVSR.py:42-60, common.py:96-109)."""
    y = lrelu(conv(x, W, name + '.0.0', stride=stride0), 0.2)
    return lrelu(conv(y, W, name + '.1.0'), 0.2)


def resblocks_with_input_conv(x, W, name, nb):
    """RefVSR.py:327-360 + mmedit sr_backbone_utils.py:42-97 (conv-ReLU-conv + identity)."""
    y = lrelu(conv(x, W, name + '.main.0'), 0.1)
    for i in range(nb):
        p = '%s.main.2.%d' % (name, i)
        y = y + conv(F.relu(conv(y, W, p + '.conv1')), W, p + '.conv2')
    return y


def pixel_shuffle_pack(x, W, name):
    """mmedit upsample.py:8-51: conv(C->4C) then pixel_shuffle(2)."""
    y = conv(x, W, name + '.upsample_conv')
    n, c4, h, w = y.shape
    c = c4 // 4
    y = y.reshape(n, c, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3)
    return y.reshape(n, c, 2 * h, 2 * w)


def spynet(ref, supp, W, prefix='Network.FlowNet'):
    """SPyNet.forward + compute_flow (models/archs/SPyNet.py:49-139)."""
    n, _, h, w = ref.shape
    w_up = w if w % 32 == 0 else 32 * (w // 32 + 1)
    h_up = h if h % 32 == 0 else 32 * (h // 32 + 1)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    r = [(resize(ref, (h_up, w_up), 'bilinear') - mean) / std]
    s = [(resize(supp, (h_up, w_up), 'bilinear') - mean) / std]
    for _ in range(5):
        r.append(avg_pool2(r[-1]))
        s.append(avg_pool2(s[-1]))
    r, s = r[::-1], s[::-1]
    flow = torch.zeros(n, 2, h_up // 32, w_up // 32)
    for lvl in range(6):
        fu = flow if lvl == 0 else flow_up2(flow)
        x = torch.cat([r[lvl], flow_warp_border(s[lvl], fu), fu], 1)
        p = '%s.basic_module.%d.basic_module' % (prefix, lvl)
        for j in range(5):
            x = conv(x, W, '%s.%d.conv' % (p, j))
            if j < 4:
                x = F.relu(x)
        flow = fu + x
    flow = resize(flow, (h, w), 'bilinear')
    scale = torch.tensor([float(w) / float(w_up), float(h) / float(h_up)]).view(1, 2, 1, 1)
    return flow * scale


def feature_extract(x, W, hd, prefix='Network.feature_match.feature_extract'):
    """VGG19[0:4]+map64 (scale 4, non-HD) or VGG19[0:7]+map128 (HD) (attention.py:28-42)."""
    y = F.relu(conv(x, W, prefix + '.0'))
    y = F.relu(conv(y, W, prefix + '.2'))
    if hd:
        y = max_pool2(y)
        y = F.relu(conv(y, W, prefix + '.5'))
        return lrelu(conv(y, W, prefix + '.map128.0'), 0.2)
    return lrelu(conv(y, W, prefix + '.map64.0'), 0.2)


def match_features(lr, ref, W, hd, scale=4):
    """The two L2-normalised patch matrices of FeatureMatching.forward (attention.py:58-85).
    Returns ref_p [n, Hr*Wr, 144], lr_p [n, 144, H*W], (Hf, Wf)."""
    lr = conv(lr, W, 'Network.feature_match.sub_mean')
    ref = conv(ref, W, 'Network.feature_match.sub_mean')
    if hd:
        f = 1.0 / (scale // 2)
        oh, ow = int(math.floor(lr.shape[-2] * f)), int(math.floor(lr.shape[-1] * f))
        lr = resize(lr, (oh, ow), 'nearest', 1.0 / f)
        ref = resize(ref, (oh, ow), 'nearest', 1.0 / f)
    vgg7 = hd or scale != 4                          # attention.py:31-35
    lr_f = feature_extract(lr, W, vgg7)
    ref_f = feature_extract(avg_pool2(ref), W, vgg7)
    lr_p = patches3x3(lr_f)
    ref_p = patches3x3(ref_f).permute(0, 2, 1)
    ref_p = ref_p / ref_p.norm(dim=2, keepdim=True).clamp_min(1e-12)
    lr_p = lr_p / lr_p.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return ref_p.contiguous(), lr_p.contiguous(), lr_f.shape[-2:]


def match_argmax(ref_p, lr_p, chunk=None):
    """max / argmax over the reference axis of ref_p @ lr_p (attention.py:91).  torch.max returns
    the first maximal index.  `chunk` splits the LR columns so the
    [Hr*Wr x H*W] matrix (16.8 GB at 270p) is never materialised; per-column results are
    unchanged."""
    n, _, L = lr_p.shape
    if chunk is None or chunk >= L:
        return torch.max(torch.bmm(ref_p, lr_p), dim=1)
    vals, idxs = [], []
    for c0 in range(0, L, chunk):
        v, i = torch.max(torch.bmm(ref_p, lr_p[:, :, c0:c0 + chunk]), dim=1)
        vals.append(v)
        idxs.append(i)
    return torch.cat(vals, 1), torch.cat(idxs, 1)


def feature_match(lr, ref, W, hd, scale=4, chunk=None):
    """FeatureMatching.forward (attention.py:58-100) -> (conf [n,1,h,w], index [n, Hf*Wf])."""
    h = lr.shape[-2]
    ref_p, lr_p, (Hf, Wf) = match_features(lr, ref, W, hd, scale)
    val, idx = match_argmax(ref_p, lr_p, chunk)
    conf = val.view(lr.shape[0], 1, Hf, Wf)
    if h / Hf != 1.0:
        conf = bicubic_scale(conf, h / Hf, clamp=True)
    return conf, idx


def aligned_conv(x, query, ref, W, name, ks):
    """AlignedConv2d.forward (RefVSR_/alignment.py:39-178).
    x [n,C,ks*h,ks*w] features to sample, query [n,3,...] LR frame (bicubic x2 inside),
    ref [n,3,...] warped RGB reference at the size of x."""
    def enc(z):   # self.conv1: conv5x5 + LReLU + ResBlock(32) + LReLU   (alignment.py:18)
        z = lrelu(conv(z, W, name + '.conv1.0'), 0.2)
        return lrelu(res_block(z, W, name + '.conv1.2'), 0.2)
    q = enc(bicubic_scale(query, 2, clamp=False))
    r = enc(ref)
    a = lrelu(conv(torch.cat([r, q], 1), W, name + '.p_conv.0', stride=ks), 0.2)
    a = lrelu(res_block(a, W, name + '.p_conv.2'), 0.2)
    affine = (conv(a, W, name + '.p_conv.4') + 1.0).clamp(-3, 3)        # [n,3,h,w]
    return aligned_sample(x, affine, ks)


def aligned_sample(x, affine, ks):
    """The affine-deformable bilinear patch sampler of AlignedConv2d (alignment.py:53-100,
    102-178), written per SURVEY appendix A4.  `rows` are what the reference calls x."""
    n, C, H2, W2 = x.shape
    h, w = affine.shape[-2:]
    ys = reflect_index(torch.arange(-1, H2 + 1), H2)
    xs = reflect_index(torch.arange(-1, W2 + 1), W2)
    xp = x[:, :, ys][:, :, :, xs]                        # reflection pad 1
    Hp, Wp = H2 + 2, W2 + 2
    half = (ks - 1) // 2
    off = torch.arange(ks, dtype=x.dtype) - half - 0.5   # o_k
    pa = off.view(ks, 1).expand(ks, ks).reshape(-1)      # row offset of sub-position n=a*ks+b
    pb = off.view(1, ks).expand(ks, ks).reshape(-1)      # col offset
    sx = affine[:, 0][..., None]                          # [n,h,w,1]
    sy = affine[:, 1][..., None]
    th = (affine[:, 2][..., None] - 1.0) * 1.0472
    px = pa.view(1, 1, 1, -1) * sx
    py = pb.view(1, 1, 1, -1) * sy
    cs, sn = torch.cos(th), torch.sin(th)
    rx = px * cs + py * (-sn)
    ry = px * sn + py * cs
    i0 = (1 + ks * torch.arange(h, dtype=x.dtype)).view(1, h, 1, 1)
    j0 = (1 + ks * torch.arange(w, dtype=x.dtype)).view(1, 1, w, 1)
    pr = rx + (half + 0.5) + i0                          # [n,h,w,N] padded-row coordinate
    pc = ry + (half + 0.5) + j0
    r0 = torch.floor(pr)
    c0 = torch.floor(pc)
    r1 = (r0 + 1).clamp(0, Hp - 1)
    c1 = (c0 + 1).clamp(0, Wp - 1)
    r0 = r0.clamp(0, Hp - 1)
    c0 = c0.clamp(0, Wp - 1)
    pr = pr.clamp(0, Hp - 1)
    pc = pc.clamp(0, Wp - 1)
    g_lt = (1 + (r0 - pr)) * (1 + (c0 - pc))
    g_rb = (1 - (r1 - pr)) * (1 - (c1 - pc))
    g_lb = (1 + (r0 - pr)) * (1 - (c1 - pc))
    g_rt = (1 - (r1 - pr)) * (1 + (c0 - pc))
    flat = xp.reshape(n, C, Hp * Wp)

    def take(rr, cc):
        lin = (rr.long() * Wp + cc.long()).reshape(n, 1, -1).expand(-1, C, -1)
        return torch.gather(flat, 2, lin).reshape(n, C, h, w, ks * ks)
    out = (g_lt[:, None] * take(r0, c0) + g_rb[:, None] * take(r1, c1) +
           g_lb[:, None] * take(r0, c1) + g_rt[:, None] * take(r1, c0))
    # (h, w, a, b) -> (h*ks + a, w*ks + b)   (alignment.py:173-178)
    out = out.reshape(n, C, h, w, ks, ks).permute(0, 1, 2, 4, 3, 5)
    return out.reshape(n, C, h * ks, w * ks)


# --------------------------------------------------------------------------------------------
# the recurrent network
# --------------------------------------------------------------------------------------------

class OracleNetwork(object):
    """Stateful restatement of models/archs/RefVSR.py:Network (forward :151-325, RAP :123-149,
    upsampler :104-119).  Executes exactly what the reference executes (no cross-window caching).
    `W` is a flat state dict with the reference key names (`Network.*`)."""

    def __init__(self, config, W, match_chunk=None, match_sample=None):
        self.cfg = config
        self.W = {k: v.detach().to(torch.float32) for k, v in W.items()}
        self.C = config.mid_channels
        self.nb = config.num_blocks
        self.hd = bool(config.flag_HD_in)
        self.scale = config.scale
        self.ks = config.matching_ksize
        self.match_chunk = match_chunk
        # timing-only mode (bench.py cpu_baseline): evaluate the matching GEMM on 1/match_sample of the
        # LR columns (results of the skipped columns are tiled copies -- numerically meaningless) and
        # accumulate the time spent so the caller can scale it back up.
        self.match_sample = match_sample
        self.match_seconds = 0.0
        self.reset_state()

    def reset_state(self):
        self.forward_feat_prop_prev = None
        self.forward_flow_prev = None
        self.forward_feat_prop_UP_prev = None
        self.forward_conf_map_prop_prev = None
        self.frame_itr_num = 0
        self.max_frame_itr_num = self.cfg.reset_branch

    def export_state(self):
        """Forward-branch state kept between calls (RefVSR.py:279-283), batch dim dropped."""
        return dict(feat=self.forward_feat_prop_prev[0], flow=self.forward_flow_prev[0],
                    feat_up=self.forward_feat_prop_UP_prev[0], conf=self.forward_conf_map_prop_prev[0],
                    frame_itr_num=self.frame_itr_num)

    def import_state(self, st):
        self.forward_feat_prop_prev = st['feat'][None].clone()
        self.forward_flow_prev = st['flow'][None].clone()
        self.forward_feat_prop_UP_prev = st['feat_up'][None].clone()
        self.forward_conf_map_prop_prev = st['conf'][None].clone()
        self.frame_itr_num = int(st['frame_itr_num'])

    # -- pieces ------------------------------------------------------------------------------
    def _feature_match(self, lr, ref):
        if not self.match_sample:
            return feature_match(lr, ref, self.W, self.hd, self.scale, self.match_chunk)
        import time
        assert not self.hd
        ref_p, lr_p, (Hf, Wf) = match_features(lr, ref, self.W, self.hd, self.scale)
        L = lr_p.shape[2]
        n = max(L // self.match_sample, 1)
        t0 = time.perf_counter()
        v, i = match_argmax(ref_p, lr_p[:, :, :n].contiguous(), self.match_chunk)
        self.match_seconds += time.perf_counter() - t0
        reps = (L + n - 1) // n
        return v.repeat(1, reps)[:, :L].view(lr.shape[0], 1, Hf, Wf), i.repeat(1, reps)[:, :L]

    def _aa(self, which, lr_like, ref_rgb, index_map, value):
        """AlignedAttention.forward (attention.py:131-159) for aa1 / aa2."""
        s = self.ks // 2 if which == 'aa1' else self.ks
        align = (which == 'aa2') or (s > 1)
        oh, ow = lr_like.shape[-2] * 2, lr_like.shape[-1] * 2
        feats = block_gather(value, index_map, s, (oh, ow))
        if not align:
            return feats
        rgb = block_gather(ref_rgb, index_map, s, (oh, ow))
        return aligned_conv(feats, lr_like, rgb, self.W, 'Network.%s.align' % which, s)

    def _rap(self, lr, ref, conf_map, conf_prop, index_map, feat_prop, feat_prop_UP,
             ref_feat_down, ref_feat):
        """AA_AF_conf_prop (RefVSR.py:123-149)."""
        W = self.W
        lr_down = bicubic_scale(lr, 0.5, clamp=True)
        aligned = self._aa('aa1', lr_down, ref, index_map, ref_feat_down)
        alpha = basic2(torch.cat([conf_prop, conf_map], 1), W, 'Network.conf_fusion')
        feat_prop = feat_prop + alpha * basic2(torch.cat([feat_prop, aligned], 1), W, 'Network.feat_fusion')
        feat_prop = res_list(feat_prop, W, 'Network.feat_decoder', 8)

        aligned_up = self._aa('aa2', lr, ref, index_map, ref_feat)
        up1 = pixel_shuffle_pack(feat_prop, W, 'Network.upsample1')
        feat_prop_UP = lrelu(conv(torch.cat([feat_prop_UP, up1], 1), W, 'Network.feat_fusion2_1.0.0'), 0.2)
        cp_up = bicubic_scale(conf_prop, 2, clamp=True)
        c_up = bicubic_scale(conf_map, 2, clamp=True)
        alpha2 = basic2(torch.cat([cp_up, c_up], 1), W, 'Network.conf_fusion2')
        feat_prop_UP = feat_prop_UP + alpha2 * basic2(torch.cat([feat_prop_UP, aligned_up], 1), W, 'Network.feat_fusion2')
        feat_prop_UP = res_list(feat_prop_UP, W, 'Network.feat_decoder2', 4)
        conf_prop = torch.maximum(conf_prop, conf_map)
        return feat_prop, feat_prop_UP, conf_prop

    def _ref_feats(self, ref):
        W = self.W
        ref_feat = res_list(basic2(ref, W, 'Network.ref_encoder1'), W, 'Network.res1', 4)
        ref_feat_down = res_list(basic2(ref_feat, W, 'Network.ref_encoder2', stride0=2), W, 'Network.res2', 4)
        return ref_feat, ref_feat_down

    def _compute_up(self, bw_up, fw_up, conf_bw, conf_fw, base):
        """compute_up (RefVSR.py:104-119)."""
        W = self.W
        cb = bicubic_scale(conf_bw, 2, clamp=True)
        cf = bicubic_scale(conf_fw, 2, clamp=True)
        cat = torch.cat([bw_up, fw_up], 1)
        out = conv(cat, W, 'Network.fusion_UP')
        alpha = basic2(torch.cat([cb, cf], 1), W, 'Network.conf_fusion_BWFW')
        out = out + alpha * basic2(cat, W, 'Network.feat_fusion_BWFW')
        out = res_list(out, W, 'Network.feat_decoder_BWFW', 4)
        if self.scale == 4:
            out = lrelu(pixel_shuffle_pack(out, W, 'Network.upsample2'), 0.1)
        out = lrelu(conv(out, W, 'Network.conv_hr'), 0.1)
        return conv(out, W, 'Network.conv_last') + base

    # -- forward -----------------------------------------------------------------------------
    def phase_a(self, lrs, refs, first_hint=False, contexts=None):
        """State-independent part of Network.forward: flows, matching and the backward branch (RefVSR.py:182-238).
        contexts: {window position: (conf_map, index_map)} matchings computed elsewhere (by this same function of the same
        frame pair: the multi-rank context exchange of shard.run_wavefront) -- used instead of recomputing them."""
        W = self.W
        n, t, c, h, w = lrs.shape
        C = self.C
        ctr = t // 2
        gradio = bool(self.cfg.EVAL.is_gradio)
        ff, bf = [], []
        for j in range(t - 1):                                               # :182-186
            ff.append(torch.zeros(n, 2, h, w) if gradio else spynet(lrs[:, j + 1], lrs[:, j], W))
        for j in range(1, t):                                                # :187-191
            bf.append(torch.zeros(n, 2, h, w) if gradio else spynet(lrs[:, j - 1], lrs[:, j], W))
        conf_maps, index_maps = [None] * t, [None] * t
        for i in range(0 if first_hint else ctr, t):                         # :196-204
            if contexts is not None and i in contexts:
                conf_maps[i], index_maps[i] = contexts[i]
            else:
                conf_maps[i], index_maps[i] = self._feature_match(lrs[:, i], refs[:, i])
        # backward branch :211-238
        feat = torch.zeros(n, C, h, w)
        feat_up = torch.zeros(n, C, 2 * h, 2 * w)
        conf = torch.zeros(n, 1, h, w)
        for i in range(t - 1, ctr - 1, -1):
            if i < t - 1:
                fl = bf[i]
                feat = warp(feat, fl)
                conf = warp(conf, fl)
                feat_up = warp(feat_up, flow_up2(fl))
            feat = resblocks_with_input_conv(torch.cat([lrs[:, i], feat], 1), W,
                                             'Network.backward_resblocks', self.nb)
            rf, rfd = self._ref_feats(refs[:, i])
            feat, feat_up, conf = self._rap(lrs[:, i], refs[:, i], conf_maps[i], conf,
                                            index_maps[i], feat, feat_up, rfd, rf)
        return dict(lrs=lrs, refs=refs, ff=ff, bf=bf, conf_maps=conf_maps, index_maps=index_maps,
                    bw_up=feat_up, conf_bw=conf)

    def phase_b1(self, pa, is_first_frame):
        """Forward-branch step only (:240-283, :292-295): the serial part of phase_b."""
        return self.phase_b(pa, is_first_frame, stop_after_forward_branch=True)

    def phase_b2(self, pa):
        """BW/FW fusion + upsampler (:288-297) of a frame whose phase_b1 has run."""
        ctr = pa['lrs'].shape[1] // 2
        base = bicubic_scale(pa['lrs'][:, ctr], self.scale, clamp=True)
        out = self._compute_up(pa['bw_up'], pa['fw_up'], pa['conf_bw'], pa['conf_fw'], base).clamp(0, 1)
        return collections.OrderedDict(result=out)

    def phase_b(self, pa, is_first_frame, is_log=False, trace=None, stop_after_forward_branch=False):
        """State-dependent rest: forward branch (:240-283), BW/FW fusion + upsampler (:288-297)."""
        W = self.W
        lrs, refs, ff, conf_maps, index_maps = pa['lrs'], pa['refs'], pa['ff'], pa['conf_maps'], pa['index_maps']
        bw_up, conf_bw = pa['bw_up'], pa['conf_bw']
        n, t, c, h, w = lrs.shape
        C = self.C
        ctr = t // 2
        if self.max_frame_itr_num is not None and self.frame_itr_num == self.max_frame_itr_num:
            is_first_frame = True                                            # :168-170
        # forward branch :240-283
        if is_first_frame:
            feat = torch.zeros(n, C, h, w)
            feat_up = torch.zeros(n, C, 2 * h, 2 * w)
            conf = torch.zeros(n, 1, h, w)
            range_start = 0                                                  # :173-176
            for i in range(0, ctr):                                          # matching of the frames before the centre
                if conf_maps[i] is None:                                     # (:196-204 with range_start = 0)
                    conf_maps[i], index_maps[i] = self._feature_match(lrs[:, i], refs[:, i])
        else:
            range_start = ctr
        fw_flow_in = self.forward_flow_prev
        for i in range(range_start, ctr + 1):
            if i > range_start:
                fl = ff[i - 1]
                feat = warp(feat, fl)
                feat_up = warp(feat, flow_up2(fl))       # :254 -- LR state resampled on the 2x grid
                conf = warp(conf, fl)
            elif not is_first_frame:
                fl = self.forward_flow_prev
                feat = warp(self.forward_feat_prop_prev, fl)
                feat_up = warp(self.forward_feat_prop_UP_prev, flow_up2(fl))
                conf = warp(self.forward_conf_map_prop_prev, fl)
            feat = resblocks_with_input_conv(torch.cat([lrs[:, i], feat], 1), W,
                                             'Network.forward_resblocks', self.nb)
            rf, rfd = self._ref_feats(refs[:, i])
            feat, feat_up, conf = self._rap(lrs[:, i], refs[:, i], conf_maps[i], conf,
                                            index_maps[i], feat, feat_up, rfd, rf)
            if i == ctr:                                                     # :279-283
                self.forward_feat_prop_prev = feat.clone()
                self.forward_flow_prev = ff[i].clone() if i < t - 1 else None
                self.forward_feat_prop_UP_prev = feat_up.clone()
                self.forward_conf_map_prop_prev = conf.clone()
        if is_first_frame:                                                   # :292-295
            self.frame_itr_num = 0
        self.frame_itr_num += 1
        if stop_after_forward_branch:      # phase_b1 (multi-GPU wavefront test): the carried state is final here
            pa['fw_up'], pa['conf_fw'] = feat_up, conf
            return pa
        base = bicubic_scale(lrs[:, ctr], self.scale, clamp=True)            # :288
        out = self._compute_up(bw_up, feat_up, conf_bw, conf, base)
        out = out.clamp(0, 1)
        outs = collections.OrderedDict()
        outs['result'] = out
        if trace is not None:
            trace.update(forward_flows=torch.stack(ff, 1), backward_flows=torch.stack(pa['bf'], 1),
                         conf_maps=conf_maps, index_maps=index_maps, backward_feat_UP=bw_up,
                         conf_map_prop_backward=conf_bw, conf_map_prop_forward=conf,
                         forward_feat_UP=feat_up, is_first_frame=is_first_frame)
        if is_log:                                                           # debugging samples, RefVSR.py:219-221,262-263,301-316
            vis = collections.OrderedDict()
            if ctr < t - 1:                                                  # :219-221 (backward loop, i == t//2)
                vis['BW_LR_next_warp'] = warp(lrs[:, ctr + 1], pa['bf'][ctr])
            # :262-263 -- `flow` is whatever the forward loop assigned last at i == t//2
            fl_fw = ff[ctr - 1] if ctr > range_start else (None if is_first_frame else fw_flow_in)
            if fl_fw is not None:
                vis['FW_LR_prev_warp'] = warp(lrs[:, ctr - 1], fl_fw)
            if self.cfg.save_sample:                                         # :301-316
                vis.update(self.sample_vis(lrs[:, ctr], refs[:, ctr], index_maps[ctr], conf_maps[ctr], conf_bw, conf))
            outs['vis'] = vis
        if is_log and self.cfg.save_sample:                                  # :318-322
            ev = collections.OrderedDict()
            ev['conf_map'] = conf_maps[ctr]
            ev['conf_map_prop'] = torch.maximum(conf_bw, conf)
            ev['conf_map_prop_backward'] = conf_bw
            ev['conf_map_prop_forward'] = conf
            outs['eval_vis'] = ev
        return outs

    def sample_vis(self, lr_c, ref_c, idx_c, conf_c, conf_bw, conf_fw):
        """The save_sample block of the debugging samples (RefVSR.py:301-316; the same code in RefVSR_IR.py:368-384)."""
        W = self.W
        vis = collections.OrderedDict()
        lr_down = bicubic_scale(lr_c, 0.5, clamp=True)
        ref_down = bicubic_scale(ref_c, 0.5, clamp=True)
        s1, s2 = self.ks // 2, self.ks
        o1 = (lr_down.shape[-2] * 2, lr_down.shape[-1] * 2)
        o2 = (lr_c.shape[-2] * 2, lr_c.shape[-1] * 2)
        fm1 = block_gather(ref_down, idx_c, s1, o1)
        vis['FW_aa1_fm_ref_aligned'] = fm1
        if s1 > 1:                                                           # aa1.align exists (HD configs)
            vis['FW_aa1_ref_aligned'] = aligned_conv(fm1, lr_down, block_gather(ref_c, idx_c, s1, o1), W, 'Network.aa1.align', s1)
        fm2 = block_gather(ref_c, idx_c, s2, o2)
        vis['FW_aa2_fm_ref_aligned'] = fm2
        vis['FW_aa2_ref_aligned'] = aligned_conv(fm2, lr_c, fm2, W, 'Network.aa2.align', s2)
        vis['conf_map_norm'] = norm_res_vis(conf_c)
        vis['conf_map_prop_backward_norm'] = norm_res_vis(conf_bw)
        vis['conf_map_prop_forward_norm'] = norm_res_vis(conf_fw)
        vis['conf_map_prop_norm'] = norm_res_vis(torch.maximum(conf_bw, conf_fw))
        return vis

    def forward(self, lrs, refs, is_first_frame, is_log=False, is_train=False, trace=None):
        """Network.forward (RefVSR.py:151-325), inference semantics (is_train=False); = phase_a + phase_b (the split
        exists for the multi-GPU wavefront test: phase A is state-free, phase B carries the forward-branch state)."""
        assert not is_train, 'oracle covers the inference path only'
        first = bool(is_first_frame) or (self.max_frame_itr_num is not None and self.frame_itr_num == self.max_frame_itr_num)
        return self.phase_b(self.phase_a(lrs, refs, first_hint=first), is_first_frame, is_log=is_log, trace=trace)

    __call__ = forward


def norm_res_vis(res):
    """models/utils.py:23-32: per-sample min/max normalisation of a debug map."""
    b = res.shape[0]
    r = res.reshape(b, -1)
    r = r - r.min(1, keepdim=True)[0]
    r = r / r.max(1, keepdim=True)[0]
    return r.view(res.shape)


def psnr(a, b):
    """trainers/trainer.py:252-254."""
    mse = torch.mean((a - b) ** 2)
    return float(10.0 * torch.log10(1.0 / mse))
