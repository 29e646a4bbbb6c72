"""Host-side packing of conv weights into the MFMA fragment order consumed by
refvsr_conv_mfma (refvsr_amd/csrc/conv_mfma.hip).

GEMM view: D[row][pixel] = sum_k Wk[row][k] * X[k][pixel].
  k-block = (tap, cg) (tap = ky*ks + kx, cg = 16-byte channel group of the concatenated, padded
  input: 8 halfs, or 4 floats in f32 mode), placed at slot kslot(ky, kx, cg) -- even-parity blocks
  ((kx + cg) & 1 == 0) first, then the odd ones, so that the two K-blocks one ds_read_b128 lane group
  mixes never collide on LDS banks (csrc/common.h:rv_kslot); K-step s covers slots 4s..4s+3; lane l of a
  wave supplies slot 4s + (l>>4) for row (l&15) of a 16-row tile.
  fp16 : packed[z][s][m][hi|lo][lane][8] = split(Wk[(z*MT+m)*16 + (lane&15)][(4s + (lane>>4))*8 : +8])
         with hi = fp16(w), lo = fp16(w - hi)  (two MFMAs per fragment, ~22-bit weights)
  f32  : packed[z][s][m][lane][4]        = Wk[...][(4s + (lane>>4))*4 : +4]
Rows are the conv output channels, except for pixel-shuffle convs where row r = sub*C + c holds
conv channel c*4 + sub so that a lane's 4 consecutive rows are 4 consecutive channels of ONE
output pixel (mmedit upsample.py:49-50 pixel_shuffle fused into the store).
"""
import numpy as np
import torch


def _pad8(c):
    return (c + 7) // 8 * 8


def _padg(c, g):
    return (c + g - 1) // g * g


def choose_mt(cout):
    n_mt = (cout + 15) // 16
    if n_mt <= 3:
        return n_mt
    if n_mt % 3 == 0:
        return 3
    if n_mt % 2 == 0:
        return 2
    return 3


def keven(ks, ncg):
    """Number of even-parity K-blocks (csrc/common.h:rv_keven)."""
    ce, co = (ncg + 1) >> 1, ncg >> 1
    return ks * (((ks + 1) >> 1) * ce + (ks >> 1) * co)


def ksteps(ks, ncg):
    return (ks * ks * ncg + (keven(ks, ncg) & 1) + 3) // 4


def kslot(ty, tx, cg, ks, ncg):
    """Slot of K-block (ty, tx, cg) in the packed K order (csrc/common.h:rv_kslot, same closed form)."""
    ce, co = (ncg + 1) >> 1, ncg >> 1
    p = (tx + cg) & 1
    c0, c1 = (co, ce) if p else (ce, co)
    row = ((ks + 1) >> 1) * c0 + (ks >> 1) * c1
    rank = ty * row + ((tx + 1) >> 1) * c0 + (tx >> 1) * c1 + (cg >> 1)
    if not p:
        return rank
    E = keven(ks, ncg)
    return E + (E & 1) + rank


def kmatrix(w, src_channels, shuffle=False, grp=8):
    """Wk [rows, G*grp] (fp32 numpy) + row->conv-channel map.  src_channels: real channel count of each
    concatenated HWC source (each is padded to a multiple of grp channels in its own buffer)."""
    w = np.asarray(w, np.float32)
    cout, cin, ks, _ = w.shape
    assert sum(src_channels) == cin, (src_channels, cin)
    pads = [_padg(c, grp) for c in src_channels]
    ncg = sum(pads) // grp
    # padded-channel index -> original input channel (or -1)
    cmap = []
    base = 0
    for c, p in zip(src_channels, pads):
        cmap += [base + i if i < c else -1 for i in range(p)]
        base += c
    cmap = np.asarray(cmap)
    rows = np.arange(cout)
    if shuffle:
        C = cout // 4
        rows = (np.arange(cout) % C) * 4 + np.arange(cout) // C      # row r -> conv channel
    G = ks * ks * ncg
    Wk = np.zeros((cout, G * grp), np.float32)
    valid = cmap >= 0
    for tap in range(ks * ks):
        ky, kx = divmod(tap, ks)
        blk = np.zeros((cout, ncg * grp), np.float32)
        blk[:, valid] = w[rows][:, cmap[valid], ky, kx]
        Wk[:, tap * ncg * grp:(tap + 1) * ncg * grp] = blk
    return Wk, rows, ncg


def pack_conv(w, b, src_channels, shuffle=False, mt=None, f32=False, hi_only=False):
    """Returns dict(wpack, bias fp32 [nz*MT*16], cout, ksteps, mt, ksize, cpads, shuffle, f32, hi_only).
    wpack: fp16 [nz,S,MT,2,64,8] (hi, lo), fp16 [nz,S,MT,1,64,8] (hi_only: plain fp16 weights, descriptor mode 2: the streamed
    kernels only -- SPyNet's 7x7 convs) or fp32 [nz,S,MT,64,4]."""
    w = w.detach().cpu().float().numpy() if isinstance(w, torch.Tensor) else np.asarray(w, np.float32)
    b = b.detach().cpu().float().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float32)
    cout, cin, ks, _ = w.shape
    grp = 4 if f32 else 8
    assert not (f32 and shuffle)
    Wk, rows, ncg = kmatrix(w, src_channels, shuffle, grp)
    S = ksteps(ks, ncg)
    MT = mt or choose_mt(cout)
    n_mt = (cout + 15) // 16
    nz = (n_mt + MT - 1) // MT
    R = nz * MT * 16
    full = np.zeros((R, S * 4 * grp), np.float32)
    for tap in range(ks * ks):
        for cg in range(ncg):
            g, j = tap * ncg + cg, kslot(tap // ks, tap % ks, cg, ks, ncg)
            full[:cout, j * grp:(j + 1) * grp] = Wk[:, g * grp:(g + 1) * grp]
    # [nz, MT, lr, S, q, grp] -> [nz, S, MT, q, lr, grp]
    frag = full.reshape(nz, MT, 16, S, 4, grp).transpose(0, 3, 1, 4, 2, 5).reshape(nz, S, MT, 64, grp)
    frag = torch.from_numpy(np.ascontiguousarray(frag))
    assert not (f32 and hi_only)
    if f32:
        wpack = frag
    elif hi_only:
        assert S > 16 and MT <= 2 and not shuffle, 'hi_only weights are for the streamed convs (more than 16 K-steps, MT <= 2)'
        wpack = frag.to(torch.float16).unsqueeze(3).contiguous()           # [nz,S,MT,1,64,8]
    else:
        hi = frag.to(torch.float16)
        lo = (frag - hi.float()).to(torch.float16)
        wpack = torch.stack([hi, lo], 3).contiguous()           # [nz,S,MT,2,64,8]
    bias = np.zeros(R, np.float32)
    bias[:cout] = b[rows]
    return dict(wpack=wpack, bias=torch.from_numpy(bias), cout=cout, ksteps=S, mt=MT, ksize=ks,
                cpads=[_padg(c, grp) for c in src_channels], shuffle=shuffle, f32=f32, hi_only=hi_only,
                raw=(torch.from_numpy(w.copy()), torch.from_numpy(b.copy())),       # fp32 originals: repacking for specialised kernels
                src_channels=list(src_channels))


# ---- the 24-channel fused residual block (csrc/resblock24.hip) ---------------------------------------------------------
RB24_S, RB24_NF, RB24_BLOB = 7, 3, 43264          # K-steps, fragments per K-step, bytes per block (REFVSR_RESBLOCK24_BLOB_BYTES)


def rb24_kblock(s, q):
    """K-block (K-step s, quarter q = lane >> 4) of the blob's K order -> (ty, tx, cg) of the 3x3 x 24-channel window, or None
    for the zero block (csrc/resblock24.hip:refvsr_resblock24_kblock is the same table; tests/test_host.py pins them together).
    The nine 16-byte slots u = 3*tx + cg of one window row are consecutive in the x tile: K-step 2*ty + a takes
    u = 4a + {0,2,1,3}[q], K-step 6 takes u = 8 of rows ty = q."""
    if s < 6:
        ty, u = s >> 1, 4 * (s & 1) + (0, 2, 1, 3)[q]
    else:
        if q == 3:
            return None
        ty, u = q, 8
    return ty, u // 3, u % 3


def pack_resblock24(w1, b1, w2, b2):
    """One fused block -> uint8 [43264]:  [conv1 fragments][conv2 fragments][b1: 32 floats][b2: 32 floats].
    Fragments of a conv: fp16 [7 K-steps][3][64 lanes][8]; lane l = (q = l >> 4, r = l & 15) holds the 8 input channels of
    K-block rb24_kblock(s, q) for row r of   f = 0: hi(W[r])   f = 1: lo(W[r])   f = 2: hi(W[16 + r]) if r < 8 else lo(W[8 + r]),
    hi = fp16(w), lo = fp16(w - hi)."""
    out = np.zeros(RB24_BLOB, np.uint8)
    o = 0
    for w in (w1, w2):
        w = w.detach().cpu().float().numpy() if isinstance(w, torch.Tensor) else np.asarray(w, np.float32)
        assert w.shape == (24, 24, 3, 3), w.shape
        hi = w.astype(np.float16)
        lo = (w - hi.astype(np.float32)).astype(np.float16)
        frag = np.zeros((RB24_S, RB24_NF, 4, 16, 8), np.float16)       # [s][f][q][r][8]
        for s_ in range(RB24_S):
            for q in range(4):
                kb = rb24_kblock(s_, q)
                if kb is None:
                    continue
                ty, tx, cg = kb
                ch = slice(cg * 8, cg * 8 + 8)
                frag[s_, 0, q] = hi[0:16, ch, ty, tx]
                frag[s_, 1, q] = lo[0:16, ch, ty, tx]
                frag[s_, 2, q, 0:8] = hi[16:24, ch, ty, tx]
                frag[s_, 2, q, 8:16] = lo[16:24, ch, ty, tx]
        raw = frag.reshape(-1).view(np.uint8)
        out[o:o + raw.size] = raw
        o += raw.size
    assert o == 2 * RB24_S * RB24_NF * 1024
    for b in (b1, b2):
        b = b.detach().cpu().float().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float32)
        bb = np.zeros(32, np.float32)
        bb[:24] = b
        out[o:o + 128] = bb.view(np.uint8)
        o += 128
    assert o == RB24_BLOB
    return torch.from_numpy(out)


# ---- 24-output-channel 3x3 convs (csrc/conv24.hip) -----------------------------------------------------------------------
def c24_steps(ncg):
    return {1: 3, 2: 5, 3: 7, 4: 9, 6: 14, 7: 18, 12: 27}[ncg]


def c24_kblock(ncg, s, q):
    """K-block (K-step s, quarter q) of the conv24 blob for ncg 16-byte channel groups of the concatenated (padded) sources
    -> (ty, tx, cg) or None for a zero block; csrc/conv24.hip:c24_kblock is the same table (tests/test_host.py pins them).
    A K-step takes four slots u = tx * (ncg | 1) + cg of the staged window whose LDS offsets are (immediate) + (one of <= 4
    per-lane patterns), the blocks of quarters (0, 1) and of (2, 3) having slot offsets of equal parity."""
    perm = (0, 2, 1, 3)
    if ncg == 1:
        return None if q == 3 else (s, perm[q], 0)
    if ncg == 3:
        if s < 6:
            u = 4 * (s & 1) + perm[q]
            return s >> 1, u // 3, u % 3
        return None if q == 3 else (q, 2, 2)
    if ncg == 4:
        return s // 3, s % 3, perm[q]
    if ncg == 2:
        if s < 3:
            u = (0, 4, 1, 3)[q]
            return s, u // 3, u % 3
        if s == 3:
            return q & 1, 2, q >> 1
        return None if q & 1 else (2, 2, q >> 1)
    if ncg == 7:                                    # two steps per tap: cg = {0, 2, 1, 3}, then {4, 6, 5, zero block}
        if s & 1:
            return None if q == 3 else (s // 6, (s // 2) % 3, 4 + perm[q])
        return s // 6, (s // 2) % 3, perm[q]
    if ncg == 12:                                   # three steps per tap: cg = 4 j + {0, 2, 1, 3}
        return s // 9, (s // 3) % 3, 4 * (s % 3) + perm[q]
    if ncg == 6:
        if s < 9:
            return s // 3, s % 3, perm[q]
        if s < 12:
            return s - 9, (0, 1, 0, 1)[q], (4, 5, 5, 4)[q]
        if s == 12:
            return q & 1, 2, 4 + (q >> 1)
        return None if q & 1 else (2, 2, 4 + (q >> 1))
    raise ValueError(ncg)


def pack_conv_hr_last(w_hr, b_hr, w_last, b_last):
    """uint8 [43264] blob for refvsr_conv_hr_last (mid_channels = 24): the resblock24 block layout with conv1 = conv_hr (24 -> 24)
    and conv2 = the output head conv_last (24 -> 3): its fragment slot (s, 0) holds hi(W_last) in rows 0-2 and lo(W_last) in rows
    8-10 of K-block rb24_kblock(s, q), slots (s, 1) and (s, 2) stay zero; b1 = conv_hr's bias, b2 = [b_last, 0, ...]."""
    w_last = w_last.detach().cpu().float().numpy() if isinstance(w_last, torch.Tensor) else np.asarray(w_last, np.float32)
    b_last = b_last.detach().cpu().float().numpy() if isinstance(b_last, torch.Tensor) else np.asarray(b_last, np.float32)
    assert tuple(w_last.shape) == (3, 24, 3, 3), w_last.shape
    out = pack_resblock24(w_hr, b_hr, np.zeros((24, 24, 3, 3), np.float32), np.zeros(24, np.float32)).numpy().copy()
    hi = w_last.astype(np.float16)
    lo = (w_last - hi.astype(np.float32)).astype(np.float16)
    frag = np.zeros((RB24_S, RB24_NF, 4, 16, 8), np.float16)
    for s_ in range(RB24_S):
        for q in range(4):
            kb = rb24_kblock(s_, q)
            if kb is None:
                continue
            ty, tx, cg = kb
            ch = slice(cg * 8, cg * 8 + 8)
            frag[s_, 0, q, 0:3] = hi[:, ch, ty, tx]
            frag[s_, 0, q, 8:11] = lo[:, ch, ty, tx]
    wb = RB24_S * RB24_NF * 1024
    out[wb:2 * wb] = frag.reshape(-1).view(np.uint8)
    bb = np.zeros(32, np.float32)
    bb[:3] = b_last
    out[2 * wb + 128:2 * wb + 256] = bb.view(np.uint8)
    return torch.from_numpy(out)


def pack_conv_last(w, b):
    """uint8 blob of the output head for refvsr_conv_last: 3x3, C -> 3 (C = 24 | 48).  [S K-steps][ONE fragment: 64 lanes x 8 halfs]
    + 32 bias floats; lane l = (q, r) of K-step s holds the 8 input channels of K-block c24_kblock(C / 8, s, q) for fragment row r:
    rows 0-2 = hi(W[r]) = fp16(w), rows 8-10 = lo(W[r - 8]) = fp16(w - hi), all other rows zero (the kernel folds rows r and r + 8)."""
    w = w.detach().cpu().float().numpy() if isinstance(w, torch.Tensor) else np.asarray(w, np.float32)
    b = b.detach().cpu().float().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float32)
    assert w.shape[0] == 3 and w.shape[1] in (24, 48) and tuple(w.shape[2:]) == (3, 3), w.shape
    Wk, _, ncg = kmatrix(w, [w.shape[1]])                       # [3, 9 * ncg * 8]
    S = c24_steps(ncg)
    hi = Wk.astype(np.float16)
    lo = (Wk - hi.astype(np.float32)).astype(np.float16)
    frag = np.zeros((S, 1, 4, 16, 8), np.float16)
    for s_ in range(S):
        for q in range(4):
            kb = c24_kblock(ncg, s_, q)
            if kb is None:
                continue
            ty, tx, cg = kb
            g = (ty * 3 + tx) * ncg + cg
            frag[s_, 0, q, 0:3] = hi[:, g * 8:g * 8 + 8]
            frag[s_, 0, q, 8:11] = lo[:, g * 8:g * 8 + 8]
    out = np.zeros(S * 1024 + 128, np.uint8)
    raw = frag.reshape(-1).view(np.uint8)
    out[:raw.size] = raw
    bb = np.zeros(32, np.float32)
    bb[:3] = b
    out[raw.size:] = bb.view(np.uint8)
    return torch.from_numpy(out)


RB48_WB = 14 * 6 * 1024                       # fragment bytes of one 48 -> 48 conv (csrc/resblock48.hip)
RB48_BLOB = 2 * RB48_WB + 512


def pack_resblock48(w1, b1, w2, b2):
    """One fused 48-channel block -> uint8 [172544] for refvsr_resblock48_chain: the fragment parts of the two refvsr_conv48 blobs
    (pack_conv24: [14 K-steps][6 fragments = hi | lo of output channels 0-15, 16-31, 32-47][64 lanes][8 halfs]) back to back, then
    b1 and b2 as 64 floats each (48.. = 0)."""
    out = np.zeros(RB48_BLOB, np.uint8)
    for i, (w, b) in enumerate(((w1, b1), (w2, b2))):
        assert tuple(w.shape) == (48, 48, 3, 3), tuple(w.shape)
        blob = pack_conv24(w, b, [48]).numpy()
        assert blob.size == RB48_WB + 256
        out[i * RB48_WB:(i + 1) * RB48_WB] = blob[:RB48_WB]
        out[2 * RB48_WB + i * 256:2 * RB48_WB + (i + 1) * 256] = blob[RB48_WB:]
    return torch.from_numpy(out)


def conv24_ok(w_shape, src_channels, shuffle=False, f32=False, shuffle_group=False, half_group=False):
    """Shapes served by the specialised kernels (csrc/conv24.hip): 3x3, 24 / 32 / 48 output channels, the listed inputs --
    the same lists as the library's refvsr_conv{24,32,48}_supported (pinned against each other in tests/test_capi.py).
    shuffle_group=True (internal, pack_conv_shuffle2 only): the 24 -> 48 row groups of the pixel-shuffle conv, which
    refvsr_conv48 itself does NOT serve (a plain 24 -> 48 conv goes to the generic kernel)."""
    cout, cin, ks, _ = w_shape
    pads = [_pad8(c) for c in src_channels]
    if ks != 3 or shuffle or f32:
        return False
    if cout == 24:
        return pads in ([24], [16], [8, 24], [24, 24]) or (half_group and pads == [48, 48])
    if cout == 32:
        return pads in ([32], [8])              # AlignedConv2d: the 32 -> 32 convs and the RGB stem
    if cout == 48:
        return pads in ([48], [16], [48, 48], [8, 48]) or (shuffle_group and pads == [24])
    return False


def conv_shuffle2_ok(w_shape, src_channels, f32=False):
    """C -> 4 C 3x3 pixel-shuffle convs served by refvsr_conv_shuffle2 (csrc/conv24.hip): C = 24 | 48."""
    cout, cin, ks, _ = w_shape
    return ks == 3 and not f32 and list(src_channels) in ([24], [48]) and cin == src_channels[0] and cout == 4 * cin


def pack_conv24(w, b, src_channels, shuffle_group=False, half_group=False):
    """uint8 blob of one conv for refvsr_conv24 / refvsr_conv48: fp16 [S][NF][64 lanes][8] + bias floats (32 | 64).  Lane
    l = (q = l >> 4, r = l & 15) of K-step s holds the 8 (padded) input channels of K-block c24_kblock(ncg, s, q) for row r of
    fragment f (hi = fp16(w), lo = fp16(w - hi)):
      24 outputs (NF = 3): f = 0: hi(W[r])   f = 1: lo(W[r])   f = 2: hi(W[16 + r]) if r < 8 else lo(W[8 + r])
      32 | 48 outputs (NF = 4 | 6): f = 2 m: hi(W[16 m + r])   f = 2 m + 1: lo(W[16 m + r])"""
    w = w.detach().cpu().float().numpy() if isinstance(w, torch.Tensor) else np.asarray(w, np.float32)
    b = b.detach().cpu().float().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float32)
    assert conv24_ok(w.shape, src_channels, shuffle_group=shuffle_group, half_group=half_group), (w.shape, src_channels)
    cout = w.shape[0]
    if cout == 48 and [_pad8(c) for c in src_channels] == [48, 48]:
        # 48 + 48 -> 48 (refvsr_conv48's two-source form): two channel-half blobs of the 24-output layout (NCG = 12 plan), back to back
        return torch.cat([pack_conv24(w[0:24], b[0:24], src_channels, half_group=True), pack_conv24(w[24:48], b[24:48], src_channels, half_group=True)])
    Wk, _, ncg = kmatrix(w, src_channels)                       # [cout, 9 * ncg * 8], K-block g = tap * ncg + cg
    S = c24_steps(ncg)
    nf = 3 if cout == 24 else cout // 8
    hi = Wk.astype(np.float16)
    lo = (Wk - hi.astype(np.float32)).astype(np.float16)
    frag = np.zeros((S, nf, 4, 16, 8), np.float16)
    for s_ in range(S):
        for q in range(4):
            kb = c24_kblock(ncg, s_, q)
            if kb is None:
                continue
            ty, tx, cg = kb
            g = (ty * 3 + tx) * ncg + cg
            cols = slice(g * 8, g * 8 + 8)
            if cout == 24:
                frag[s_, 0, q] = hi[0:16, cols]
                frag[s_, 1, q] = lo[0:16, cols]
                frag[s_, 2, q, 0:8] = hi[16:24, cols]
                frag[s_, 2, q, 8:16] = lo[16:24, cols]
            else:
                for m in range(cout // 16):
                    frag[s_, 2 * m, q] = hi[16 * m:16 * m + 16, cols]
                    frag[s_, 2 * m + 1, q] = lo[16 * m:16 * m + 16, cols]
    nb = 64 if cout == 48 else 32
    out = np.zeros(S * nf * 1024 + nb * 4, np.uint8)
    raw = frag.reshape(-1).view(np.uint8)
    out[:raw.size] = raw
    bb = np.zeros(nb, np.float32)
    bb[:cout] = b
    out[raw.size:] = bb.view(np.uint8)
    return torch.from_numpy(out)


def pack_conv_shuffle2(w, b):
    """uint8 blobs of the C -> 4 C pixel-shuffle conv for refvsr_conv_shuffle2 (C = 24 | 48): 2 | 4 conv48-style blobs of 48 output
    rows, back to back.  F.pixel_shuffle(2) puts conv channel 4 c + 2 dy + dx at channel c of sub-pixel (dy, dx); the kernel wants
    a lane's four consecutive accumulator rows to be four consecutive channels of one sub-pixel:
      C = 24: blob z = dy, row R -> sub-pixel dx = R // 24, channel R % 24;   C = 48: blob z = 2 dy + dx, row R = channel R."""
    w = w.detach().cpu().float() if isinstance(w, torch.Tensor) else torch.from_numpy(np.asarray(w, np.float32))
    b = b.detach().cpu().float() if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b, np.float32))
    c = w.shape[1]
    assert conv_shuffle2_ok(tuple(w.shape), [c]), tuple(w.shape)
    blobs = []
    for z in range(2 if c == 24 else 4):
        R = np.arange(48)
        rows = 4 * (R % 24) + 2 * z + R // 24 if c == 24 else 4 * R + z
        blobs.append(pack_conv24(w[rows], b[rows], [c], shuffle_group=True))
    return torch.cat(blobs)
