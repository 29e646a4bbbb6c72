// 3x3 stride-1 convolutions with 24 output channels on fp16 HWC maps, everything a compile-time constant: the single-conv
// companion of resblock24.hip (same design: K order with immediate-offset B-fragment reads, hi + lo weights in three 16-row
// fragments with a half-wave fold, bias as the accumulators' initial value, one parameter blob per conv via LDS-DMA).
// Covers the conv shapes of the RefVSR_small family (mid_channels = 24) around the fused blocks:
//
//   NCG0 + NCG1 (16-byte channel groups of source 0 + source 1)
//   3 + 0   24 -> 24        ResList.conv_tail (RefVSR_/common.py:80-82), feat_fusion*.1 (RefVSR.py:53-62), ref_encoder*, conv_hr
//   2 + 0   16 -> 24        conf_fusion*.1 (RefVSR.py:47-52)
//   1 + 3   8 + 24 -> 24    ResidualBlocksWithInputConv.main.0 on cat([lr, feat]) (RefVSR.py:340-343)
//   3 + 3   24 + 24 -> 24   feat_fusion*.0 / feat_fusion2_1 / fusion_UP on cat([a, b]) (RefVSR.py:53-62,87)
//
// out = post( act(conv + bias) * mul + res ), the epilogue of refvsr_conv_mfma's fp16 HWC mode.  The generic kernel stays
// for every other shape (strides, 5x5 / 7x7, pixel shuffle, planar outputs, other channel counts).
//
// COUT = 32 (AlignedConv2d's 32-channel convs, RefVSR_/alignment.py:18-24): four fragments per K-step ([hi | lo] of output channels
// 0-15, 16-31), inputs 32 (NCG 4) and 8 (NCG 1: the RGB stem; K plan: the three taps of a window row + a zero block per step).
//
// COUT = 48 (the mid_channels = 48 family, configs/config_RefVSR_{L1,MFID,MFID_8K}.py: 30 ResidualBlockNoBN per branch, two
// convs each): six fragments per K-step ([hi | lo] of output channels 0-15, 16-31, 32-47; no fold), 48 -> 48 with the 84 KB
// weight set resident next to a 16 x 32-pixel tile (18 x 34 staged pixels, 68 KB) walked by SIXTEEN waves -- one workgroup per
// CU, four waves per SIMD, 255 tiles for the 270 x 480 map of the 256-CU chip -- and 16 -> 48 on the 8 x 32 tile.
//
// SHUF = 24 | 48 (PixelShufflePack.upsample_conv + F.pixel_shuffle, mmedit upsample.py:36-51; RefVSR.py:89-90,138): the
// C -> 4 C conv as groups of 48 output rows on the COUT = 48 kernel, blockIdx.y = group z, rows ordered sub-pixel-major so that
// a lane's four accumulator rows are four consecutive channels of ONE sub-pixel: C = 24: group z = output row parity dy, rows
// [dx = 0: channels 0-23][dx = 1: channels 0-23]; C = 48: group z = sub-pixel 2 dy + dx, rows = channels 0-47.  The epilogue
// stores straight into the [2h][2w][C] map (bias + optional leaky activation; no multiplier / residual on this layer).
//
// LDS: [fragments: S K-steps x 3 x 1 KiB][bias: 32 floats][x tile: 10 x 34 pixels x PS slots of 16 bytes, PS = NCG | 1 (odd
// pixel stride: bank-conflict-free B reads with the pixel permutation of common.h)].  K plans (c24_kblock): the K-blocks of
// one window row are the slots u = tx * PS + cg; a K-step takes four of them whose offsets are (step immediate) + (one of
// <= 4 per-lane patterns), the two blocks a ds_read_b128 lane group mixes having slot offsets of equal parity:
//   NCG 3: rows of 9 consecutive slots -> 2 steps of {0,2,1,3} + 4a per row, then u = 8 of the three rows   (7 steps)
//   NCG 4: one step per tap, cg = {0,2,1,3}                                                                  (9 steps)
//   NCG 2: u = {0,4,1,3} per row, then u = 6 | 7 of rows 0,1, then of row 2                                  (5 steps)
//   NCG 6: one step per tap with cg = {0,2,1,3}; cg 4,5 of taps (tx 0, tx 1) per row; of tx 2 for rows 0,1; for row 2   (14 steps)
#include "common.h"
#include "conv24_plan.h"
#include <stdlib.h>

struct C24Args {
    const unsigned char* src0; const unsigned char* src1; unsigned char* out;
    const unsigned char* blob; const unsigned char* mul; const unsigned char* res;
    int h, w, tiles_x, n_tiles, grid;
    float act_slope, post_slope;
    // CONF variants (refvsr_conf_alpha): the 16-channel input map is not read, it is COMPUTED while the tile is staged
    const float* conf_a; const float* conf_b;    // the two planar fp32 confidence maps [ch][cw]
    const float* cw0; const float* cb0;          // first conv of the pair: fp32 weights [16][2][3][3], bias [16]
    float* conf_max;                             // optional by-product max(conf_a, conf_b) [ch][cw] (CONF = 1)
    int ch, cw;                                  // size of the confidence maps (CONF = 2: half the conv's grid)
    float slope0;
    // COUT = 3 (refvsr_conv_last): `out` is planar fp32 [3][h][w]; base_lr: the LR centre frame, planar fp32 [3][bh][bw], whose
    // bicubic up-sampling (clamped to [0, 1]) is added before the final clamp
    const float* base_lr; int bh, bw; float base_step;
    int out_fmt;                                 // COUT = 3: REFVSR_RESULT_* of `out`
    // Multi-map launches (refvsr_*_batch, ABI 11): batch > 1 maps of one geometry share the launch and the weight fill; flat tile
    // index t = b * tpm + (tile of map b); map b's operands come from the tables (entry 0 == the scalar fields above, which stay
    // the "operand present" flags).  Not for the HALF variant.
    int batch, tpm;
    const unsigned char* bsrc0[REFVSR_MAX_MAPS]; const unsigned char* bsrc1[REFVSR_MAX_MAPS]; unsigned char* bout[REFVSR_MAX_MAPS];
    const unsigned char* bmul[REFVSR_MAX_MAPS]; const unsigned char* bres[REFVSR_MAX_MAPS];
    const float* bconf_a[REFVSR_MAX_MAPS]; const float* bconf_b[REFVSR_MAX_MAPS]; float* bconf_max[REFVSR_MAX_MAPS];
};

__device__ __forceinline__ float c24_fold1(const float a) {       // lane l: a[l] + a[l ^ 32] (see resblock24.hip:rb_fold1)
    const unsigned u = __float_as_uint(a);
    const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned x0 = pr[0], x1 = pr[1];
    return __uint_as_float(x0) + __uint_as_float(x1);
}

// COUT = 24 | 48 output channels; TH = 8 | 16 tile rows; NWV waves walk the TH x 32 tile, T = 2 TH / NWV pixel groups per wave;
// WPS = waves per SIMD the kernel is built for (register budget).
// CONF = 1 | 2 (16 -> COUT convs of the confidence fusions, RefVSR.py:47-52,130,141-142,107-109): the kernel's 16-channel
// input  a = lrelu(conv3x3_{2->16}(cat[conf_a, conf_b]))  -- for CONF = 2 of the pair up-sampled x2 (bicubic, clamped to [0, 1]) --
// is evaluated per staged pixel from the two fp32 confidence maps instead of being read: torch.cat, [F.interpolate,] the
// 2 -> 16 conv (refvsr_conv_direct_f32's fp32 FMA order, fp16 rounding) and for CONF = 1 the torch.max of the two maps
// (RefVSR.py:147) leave the launch list; results are bit-identical to the separate launches.
// HALF = 1 (48 + 48 -> 48, the two-source convs of the mid_channels = 48 models: feat_fusion*.0 / feat_fusion2_1 / fusion_UP on
// cat([a, b]), RefVSR.py:53-62,87): one conv's hi + lo weights are 166 KB -- no resident form.  The OUTPUT channels are split
// instead: blockIdx.y = z computes channels 24 z .. 24 z + 23 with the 24-row fragment trick (81 KB of weights per half, NCG = 12
// plan, 70 KB tile: one sixteen-wave workgroup per CU) and stores them into the 48-channel maps at byte offset 48 z.
// MM = 1 (ABI 11): multi-map launch -- p.batch maps behind one weight fill (the batched entry points; the single-map instantiations
// are untouched by it).
template <int COUT, int NCG0, int NCG1, int NWV, int TH, int WPS, int SHUF = 0, int CONF = 0, int HALF = 0, int MM = 0>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void conv24_kernel(C24Args p) {
    static_assert(HALF == 0 || (COUT == 24 && SHUF == 0 && CONF == 0), "channel-half variant: 24 computed channels per workgroup");
    static_assert(MM == 0 || (HALF == 0 && COUT != 3), "multi-map variant");
    static_assert(SHUF == 0 || (COUT == 48 && (SHUF == 24 || SHUF == 48) && NCG0 * 8 == SHUF && NCG1 == 0), "pixel-shuffle variant");
    static_assert(CONF == 0 || (NCG0 == 2 && NCG1 == 0 && SHUF == 0), "confidence variant: 16-channel single source");
    constexpr int NCG = NCG0 + NCG1, PS = NCG | 1, PXB = PS * 16, ROWB = C24_XW * PXB;
    constexpr int S = c24_steps(NCG), NPAT = c24_npat(NCG);
    constexpr int NF = COUT == 24 ? 3 : COUT == 3 ? 1 : COUT / 8;   // fragments per K-step (COUT = 32 | 48: [hi | lo] per 16 channels)
    constexpr int NM = COUT == 24 ? 2 : COUT == 3 ? 1 : COUT / 16;  // accumulator tiles per pixel group
    constexpr int BIASB = COUT == 48 ? 256 : 128;
    constexpr int WB = S * NF * 1024, BIAS = WB, XT = WB + BIASB;
    constexpr int NT = NWV * 64, T = 2 * TH / NWV;
    constexpr int C24_TH = TH, C24_XH = TH + 2, C24_NPX = C24_XH * C24_XW;
    constexpr int NCH = C24_NPX * NCG, KCH = (NCH + NT - 1) / NT;
    constexpr int PIXB0 = NCG0 * 16, PIXB1 = NCG1 * 16;
    constexpr int OPX = HALF ? 96 : COUT * 2;                       // bytes per pixel of the out / mul / res maps
    static_assert((COUT == 3 || COUT == 24 || COUT == 32 || COUT == 48) && S > 0 && T >= 1 && T * NWV == 2 * TH, "unsupported shape");
    static_assert(COUT != 3 || (NCG1 == 0 && SHUF == 0 && CONF == 0), "output head: single source");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    asm volatile("" :: "s"(p.src0), "s"(p.src1), "s"(p.out), "s"(p.blob), "s"(p.mul), "s"(p.res), "s"(p.h), "s"(p.w), "s"(p.tiles_x),
                 "s"(p.n_tiles), "s"(p.grid), "s"(p.act_slope), "s"(p.post_slope));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weights + bias: global -> LDS (1 KiB per wave instruction), issued first
    {
        constexpr int NPC = WB / 1024;                               // full pieces; the 128-byte bias tail: 8 lanes
        const unsigned char* g = p.blob + ((SHUF || HALF) ? (int)blockIdx.y * (WB + BIASB) : 0) + lane * 16;   // SHUF / HALF: one blob per row group
#pragma unroll
        for (int j = 0; j < (NPC + NWV - 1) / NWV; ++j) {
            const int c = wave + j * NWV;
            if (c < NPC)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + c * 1024),
                                                 (__attribute__((address_space(3))) void*)(smem + c * 1024), 16, 0, 0);
        }
        if (wave == NPC % NWV && lane < BIASB / 16)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + NPC * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + NPC * 1024), 16, 0, 0);
    }

    if constexpr (HALF != 0) {                                       // this workgroup's 24 channels inside the 48-channel maps
        const int zoff = (int)blockIdx.y * 48;
        p.out += zoff;
        if (p.mul) p.mul += zoff;
        if (p.res) p.res += zoff;
    }
    // ---- x-tile chunks of this thread: i = tid + k NT = pixel * NCG + cg; global offset relative to the tile origin of its
    //      source, LDS offset, source flag -- computed once
    const int rowp = p.w;
    unsigned xg[KCH], xl[KCH];
    unsigned xs1 = 0;                                                // bit k: chunk k comes from source 1
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
        const int i = min(tid + k * NT, NCH - 1);
        const int px = i / NCG, cg = i - px * NCG;
        const int r = px / C24_XW, c = px - r * C24_XW;
        const bool s1 = cg >= NCG0;
        xg[k] = (unsigned)((r * rowp + c) * (s1 ? PIXB1 : PIXB0) + (s1 ? cg - NCG0 : cg) * 16);
        xl[k] = (unsigned)(XT + px * PXB + cg * 16);
        xs1 |= s1 ? (1u << k) : 0u;
    }
    uint4 xv[KCH];
    auto x_fetch = [&](const int tf) {
        int t = tf;
        const unsigned char* s0p = p.src0;
        const unsigned char* s1p = p.src1;
        if constexpr (MM != 0) {                                     // flat tile index -> (map, tile of the map); uniform
            const int bm = (int)((unsigned)tf / (unsigned)p.tpm);
            t = tf - bm * p.tpm;
            s0p = p.bsrc0[bm];
            if constexpr (NCG1 != 0) s1p = p.bsrc1[bm];
        }
        const int tyi = t / p.tiles_x;
        const int ty0 = tyi * C24_TH, tx0 = (t - tyi * p.tiles_x) * C24_TW;
        const bool interior = ty0 >= 1 && ty0 + C24_TH + 1 <= p.h && tx0 >= 1 && tx0 + C24_TW + 1 <= p.w;
        const long long org = (long long)(ty0 - 1) * p.w + (tx0 - 1);          // origin pixel (may lie outside the frame)
        const unsigned char* b0 = s0p + org * PIXB0;
        const unsigned char* b1 = NCG1 ? s1p + org * PIXB1 : b0;
        if (interior) {
#pragma unroll
            for (int k = 0; k < KCH; ++k)
                xv[k] = *reinterpret_cast<const uint4*>((NCG1 && ((xs1 >> k) & 1) ? b1 : b0) + xg[k]);
        } else {
            int tide = tid;                                           // opaque: keeps this path's index math inside the branch
            asm volatile("" : "+v"(tide));
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                const int i = min(tide + k * NT, NCH - 1);
                const int px = i / NCG, cg = i - px * NCG;
                const int r = px / C24_XW, c = px - r * C24_XW;
                const int iy = ty0 - 1 + r, ix = tx0 - 1 + c;
                const bool ok = (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w;
                const bool s1 = cg >= NCG0;
                const unsigned pix = (unsigned)(min(max(iy, 0), p.h - 1) * p.w + min(max(ix, 0), p.w - 1));
                const unsigned char* g = s1 ? s1p + pix * PIXB1 + (cg - NCG0) * 16 : s0p + pix * PIXB0 + cg * 16;
                uint4 v = *reinterpret_cast<const uint4*>(g);        // clamped address, masked value (32-bit offsets: host check)
                const unsigned keep = ok ? 0xffffffffu : 0u;
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                xv[k] = v;
            }
        }
    };
    auto x_park = [&]() {
#pragma unroll
        for (int k = 0; k < KCH; ++k)
            if (k * NT + NT <= NCH || tid + k * NT < NCH) *reinterpret_cast<uint4*>(smem + xl[k]) = xv[k];
    };
    // ---- CONF: LDS behind the x tile: [pair tile: 2 x (TH + 4) x 36 floats][first conv: 18 taps x 16 outputs][its bias: 16]
    constexpr int PTH = TH + 4, PTW = C24_XW + 2;
    constexpr int PT = XT + C24_NPX * PXB, CW0 = PT + 2 * PTH * PTW * 4, CB0 = CW0 + 18 * 16 * 4;
    if constexpr (CONF != 0) {
        float* wl0 = reinterpret_cast<float*>(smem + CW0);            // [tap = ci * 9 + ky * 3 + kx][co]: 64-byte rows, co fastest
        for (int i = tid; i < 18 * 16; i += NT) {
            const int co = i & 15, tap = i >> 4;
            wl0[i] = p.cw0[(co * 2 + tap / 9) * 9 + tap % 9];
        }
        if (tid < 16) reinterpret_cast<float*>(smem + CB0)[tid] = p.cb0[tid];
    }
    // the whole x tile of tile t, computed: stage 1 = the (TH + 4) x 36 window of the pair (zero outside the frame = the first
    // conv's padding; CONF = 2: clamp01(bicubic x2) of the half-size maps), stage 2 = lrelu(conv 2 -> 16) per staged pixel
    // (zero outside the frame = the second conv's padding).  Called by all threads; contains one barrier.
    auto conf_stage = [&](const int tf) {
        int t = tf;
        const float* cap = p.conf_a;
        const float* cbp = p.conf_b;
        float* cmp_ = p.conf_max;
        if constexpr (MM != 0) {
            const int bm = (int)((unsigned)tf / (unsigned)p.tpm);
            t = tf - bm * p.tpm;
            cap = p.bconf_a[bm]; cbp = p.bconf_b[bm];
            if (p.conf_max) cmp_ = p.bconf_max[bm];
        }
        const int tyi = t / p.tiles_x;
        const int ty0 = tyi * C24_TH, tx0 = (t - tyi * p.tiles_x) * C24_TW;
        float* pt = reinterpret_cast<float*>(smem + PT);
        for (int i = tid; i < 2 * PTH * PTW; i += NT) {
            const int c = i / (PTH * PTW), rem = i - c * (PTH * PTW);
            const int r = rem / PTW, cc = rem - r * PTW;
            const int y = ty0 - 2 + r, x = tx0 - 2 + cc;
            float v = 0.0f;
            if ((unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w) {
                const float* src = c ? cbp : cap;
                if constexpr (CONF == 1) v = src[(size_t)y * p.cw + x];
                else v = fminf(fmaxf(rv_bicubic_at(src, p.ch, p.cw, y, x, 0.5f, 0.5f), 0.0f), 1.0f);
            }
            pt[i] = v;
        }
        __syncthreads();
        const float* wl0 = reinterpret_cast<const float*>(smem + CW0);
        const float* bl0 = reinterpret_cast<const float*>(smem + CB0);
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            const int i = tid + k * NT;
            if (i < NCH) {
                const int px = i >> 1, cg = i & 1;
                const int r = px / C24_XW, c = px - r * C24_XW;
                const int iy = ty0 - 1 + r, ix = tx0 - 1 + c;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if ((unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w) {
                    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {             // conv_direct.hip's order: ci, ky, kx; acc = fma(x, w, acc)
                                const float xv = pt[ci * (PTH * PTW) + (r + ky) * PTW + (c + kx)];
                                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl0 + (ci * 9 + ky * 3 + kx) * 16 + cg * 8);
                                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wl0 + (ci * 9 + ky * 3 + kx) * 16 + cg * 8 + 4);
                                acc[0] = fmaf(xv, w0[0], acc[0]); acc[1] = fmaf(xv, w0[1], acc[1]);
                                acc[2] = fmaf(xv, w0[2], acc[2]); acc[3] = fmaf(xv, w0[3], acc[3]);
                                acc[4] = fmaf(xv, w1[0], acc[4]); acc[5] = fmaf(xv, w1[1], acc[5]);
                                acc[6] = fmaf(xv, w1[2], acc[6]); acc[7] = fmaf(xv, w1[3], acc[7]);
                            }
                    union { f16x8 h; uint4 u; } o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o.h[j] = (f16)rv_lrelu(acc[j] + bl0[cg * 8 + j], p.slope0);
                    v = o.u;
                }
                *reinterpret_cast<uint4*>(smem + XT + px * PXB + cg * 16) = v;
            }
        }
        if constexpr (CONF == 1) {
            if (p.conf_max) {                                          // RefVSR.py:147 for the pixels this tile owns
                for (int i = tid; i < C24_TH * C24_TW; i += NT) {
                    const int r = i / C24_TW, c = i - r * C24_TW;
                    const int y = ty0 + r, x = tx0 + c;
                    if (y < p.h && x < p.w)
                        cmp_[(size_t)y * p.w + x] = fmaxf(pt[(r + 2) * PTW + c + 2], pt[PTH * PTW + (r + 2) * PTW + c + 2]);
                }
            }
        }
    };
    int tl, k_hi;
    rv_tile_range(p.n_tiles, p.grid, tl, k_hi);
    if constexpr (CONF == 0) { if (tl < k_hi) x_fetch(tl); }
    __builtin_amdgcn_sched_barrier(0);                               // weights and first tile in flight before the rest of the set-up

    // ---- per-lane constants
    const int q = lane >> 4;
    const int lp = rv_pix16(lane & 15);
    const int la = lane * 16;
    // pixel group t of this wave = group wave * T + t of the tile: output row RW(t), left | right half CG(t) (T odd: NWV = 2 TH)
    const int g0w = wave * T;
#define RW(t) ((g0w + (t)) >> 1)
#define CG(t) ((g0w + (t)) & 1)
    auto sel4 = [&](const int v0, const int v1, const int v2, const int v3) { return q == 0 ? v0 : q == 1 ? v1 : q == 2 ? v2 : v3; };
    int pb[T];                                                       // window origin of each output group + this lane's pattern-0 offset
#pragma unroll
    for (int t = 0; t < T; ++t)
        pb[t] = XT + RW(t) * ROWB + (CG(t) * 16 + lp) * PXB +
                sel4(c24_off(NCG, 0, 0), c24_off(NCG, 0, 1), c24_off(NCG, 0, 2), c24_off(NCG, 0, 3)) - c24_off(NCG, 0, 0);
    int pd[NPAT];                                                    // pattern p relative to pattern 0, per lane
#pragma unroll
    for (int pp = 0; pp < NPAT; ++pp) {
        const int s = c24_pat_step(NCG, pp);
        pd[pp] = (sel4(c24_off(NCG, s, 0), c24_off(NCG, s, 1), c24_off(NCG, s, 2), c24_off(NCG, s, 3)) - c24_off(NCG, s, 0)) -
                 (sel4(c24_off(NCG, 0, 0), c24_off(NCG, 0, 1), c24_off(NCG, 0, 2), c24_off(NCG, 0, 3)) - c24_off(NCG, 0, 0));
    }
    const int rowb_o = p.w * OPX;
    const unsigned oo = (unsigned)(lp * OPX + q * 8);                // this lane's channels 4q.. of its pixel, relative to the group's origin

    if constexpr (CONF == 0) { if (tl < k_hi) x_park(); }
    __syncthreads();                                                 // weights, bias, first tile (CONF: first-conv weights)

    for (; tl < k_hi; ++tl) {
        const bool has_next = tl + 1 < k_hi;
        if constexpr (CONF != 0) {                                   // the tile is computed, not prefetched: staged at the top
            conf_stage(tl);
            __syncthreads();
        }
        int tm = tl;
        unsigned char* outp = p.out;
        const unsigned char* mulp = p.mul;
        const unsigned char* resp = p.res;
        if constexpr (MM != 0) {
            const int bm = (int)((unsigned)tl / (unsigned)p.tpm);
            tm = tl - bm * p.tpm;
            outp = p.bout[bm];
            if (p.mul) mulp = p.bmul[bm];
            if (p.res) resp = p.bres[bm];
        }
        const int tyi = tm / p.tiles_x;
        const int ty0 = tyi * C24_TH, tx0 = (tm - tyi * p.tiles_x) * C24_TW;
        const bool interior = ty0 >= 1 && ty0 + C24_TH + 1 <= p.h && tx0 >= 1 && tx0 + C24_TW + 1 <= p.w;
        const long long oorg = ((long long)ty0 * p.w + tx0) * OPX;
        // epilogue operands of this tile, in flight during the K loop.  Accumulator tile m of a pixel group holds channels
        // 16 m + 4 q .. (COUT = 24: tile 1 = channels 16 + 4 q for q < 2, see resblock24.hip)
        f16x4 mv[NM][T], rv[NM][T];
        bool okt[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            okt[t] = true;
            if (!interior) {
                int lpe = lp;
                asm volatile("" : "+v"(lpe));
                okt[t] = ty0 + RW(t) < p.h && tx0 + CG(t) * 16 + lpe < p.w;
            }
            const unsigned eo = (unsigned)(RW(t) * rowb_o) + oo + CG(t) * 16 * OPX;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                mv[m][t] = rv[m][t] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                const bool lane_ok = okt[t] && (COUT != 24 || m == 0 || q < 2);
                if constexpr (COUT == 24) {                          // (COUT = 48: fetched in the epilogue -- register budget)
                    if (p.mul && lane_ok) mv[m][t] = *reinterpret_cast<const f16x4*>(mulp + oorg + eo + 32 * m);
                    if (p.res && lane_ok) rv[m][t] = *reinterpret_cast<const f16x4*>(resp + oorg + eo + 32 * m);
                }
                (void)lane_ok;
            }
        }
        if constexpr (CONF == 0) { if (has_next) x_fetch(tl + 1); }  // next tile: in flight during the K loop

        // ---------------- K loop: acc = bias + conv(x) on the TH x 32 tile ---------------------------------------------------
        f32x4 acc[NM][T];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + BIAS + m * 64 + q * 16);   // channels 16 m + 4 q .. (pads: zeros)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[m][t] = bv;
        }
        {
            uint4 fa[2][NF], fb[2][T];
            auto load = [&](auto sc, uint4 (&af)[NF], uint4 (&bf)[T]) {
                constexpr int s = decltype(sc)::value;
                constexpr int pp = c24_pat(NCG, s);
                constexpr int imm = c24_off(NCG, s, 0);
#pragma unroll
                for (int f = 0; f < NF; ++f) af[f] = *reinterpret_cast<const uint4*>(smem + (s * NF + f) * 1024 + la);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if constexpr (pp == 0) bf[t] = *reinterpret_cast<const uint4*>(smem + pb[t] + imm);
                    else bf[t] = *reinterpret_cast<const uint4*>(smem + pb[t] + pd[pp] + imm);
                }
            };
            auto mfma = [&](const uint4 (&af)[NF], const uint4 (&bf)[T]) {
                if constexpr (COUT == 24) {                          // [hi 0-15] [lo 0-15] [hi 16-23 | lo 16-23]
                    const f16x8 a_hi = *reinterpret_cast<const f16x8*>(&af[0]);
                    const f16x8 a_lo = *reinterpret_cast<const f16x8*>(&af[1]);
                    const f16x8 a_mx = *reinterpret_cast<const f16x8*>(&af[2]);
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, *reinterpret_cast<const f16x8*>(&bf[t]), acc[0][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_mx, *reinterpret_cast<const f16x8*>(&bf[t]), acc[1][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, *reinterpret_cast<const f16x8*>(&bf[t]), acc[0][t], 0, 0, 0);
                } else if constexpr (COUT == 3) {                    // ONE fragment: rows 0-2 = hi, rows 8-10 = lo (folded in the epilogue)
                    const f16x8 a_w = *reinterpret_cast<const f16x8*>(&af[0]);
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_w, *reinterpret_cast<const f16x8*>(&bf[t]), acc[0][t], 0, 0, 0);
                } else {                                             // [hi | lo] of channels 0-15, 16-31, 32-47: all hi, then all lo
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int m = 0; m < NM; ++m) {
                            const f16x8 av = *reinterpret_cast<const f16x8*>(&af[2 * m + h]);
#pragma unroll
                            for (int t = 0; t < T; ++t)
                                acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, *reinterpret_cast<const f16x8*>(&bf[t]), acc[m][t], 0, 0, 0);
                        }
                }
            };
            if constexpr (NF <= 4) {
                // two fragment sets: the reads of step s + 1 are issued above the MFMAs of step s
                load(std::integral_constant<int, 0>{}, fa[0], fb[0]);
                c24_static_for([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (s + 1 < S) load(std::integral_constant<int, s + 1>{}, fa[(s + 1) & 1], fb[(s + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma(fa[s & 1], fb[s & 1]);
                }, std::make_integer_sequence<int, S>{});
            } else {
                // ONE fragment set (six weight fragments: a second set does not fit 128 VGPRs): the reads of step s + 1 are issued
                // right behind the 6 T MFMAs of step s -- those have latched their operands by then and occupy the matrix pipe for
                // 96 T cycles, longer than the reads take
                load(std::integral_constant<int, 0>{}, fa[0], fb[0]);
                c24_static_for([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    mfma(fa[0], fb[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (s + 1 < S) load(std::integral_constant<int, s + 1>{}, fa[0], fb[0]);
                }, std::make_integer_sequence<int, S>{});
            }
        }
        if (has_next) {
            __syncthreads();                                         // every wave is done reading the x tile
            if constexpr (CONF == 0) x_park();
        }
        if constexpr (COUT == 3) {
            // ---------------- output head (RefVSR.py:118,288,297): clamp( conv + bias + clamp01(bicubic(lr_centre)), 0, 1 ) -> planar fp32.
            // After the fold lane (0, n) holds the three channel sums of pixel n of the group; lane (q, n), q < 3, takes channel q
            // (ds_bpermute), evaluates ITS channel's bicubic sample (rv_bicubic_at: resize_kernel<BICUBIC>'s FMA chains) and stores one
            // value: 48 lanes x 4 bytes = three 64-byte row segments per group
            const size_t plane_o = (size_t)p.h * p.w, plane_b = (size_t)p.bh * p.bw;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 y = acc[0][t];
                const float s0 = c24_fold1(y[0]), s1 = c24_fold1(y[1]), s2 = c24_fold1(y[2]);
                const int srcl = lane & 15;
                const float v0 = __shfl(s0, srcl), v1 = __shfl(s1, srcl), v2 = __shfl(s2, srcl);
                const float v = q == 0 ? v0 : q == 1 ? v1 : v2;
                int lpe = lp;
                if (!interior) asm volatile("" : "+v"(lpe));
                const int oy = ty0 + RW(t), ox = tx0 + CG(t) * 16 + lpe;
                if (q < 3 && oy < p.h && ox < p.w) {
                    const float b = fminf(fmaxf(rv_bicubic_at(p.base_lr + q * plane_b, p.bh, p.bw, oy, ox, p.base_step, p.base_step), 0.0f), 1.0f);
                    rv_store_result(outp, q * plane_o + (size_t)oy * p.w + ox, fminf(fmaxf(v + b, 0.0f), 1.0f), p.out_fmt);
                }
            }
        } else if constexpr (SHUF != 0) {
            // ---------------- pixel-shuffle epilogue: rows 16 m + 4 q .. of group z -> channels ch0 .. of sub-pixel (dy, dx) -------
            const int z = (int)blockIdx.y;
            const unsigned w2 = 2u * (unsigned)p.w;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                int lpe = lp;
                if (!interior) asm volatile("" : "+v"(lpe));
                const unsigned oy = (unsigned)(ty0 + RW(t)), ox = (unsigned)(tx0 + CG(t) * 16 + lpe);
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int r0 = 16 * m + 4 * q;
                    const int dx = SHUF == 24 ? (r0 >= 24 ? 1 : 0) : (z & 1);
                    const int dy = SHUF == 24 ? z : (z >> 1);
                    const int ch0 = SHUF == 24 ? r0 - 24 * dx : r0;
                    f32x4 y = acc[m][t];
                    if (p.act_slope != 1.0f) {                       // (the activation commutes with the shuffle: RefVSR.py:116)
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[i] = fmaxf(y[i], y[i] * p.act_slope);
                    }
                    const f16x4 o = {(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
                    if (okt[t])
                        *reinterpret_cast<f16x4*>(outp + ((2u * oy + dy) * w2 + 2u * ox + dx) * (unsigned)(SHUF * 2) + ch0 * 2) = o;
                }
            }
        } else
        // ---------------- epilogue: out = post(act(acc) * mul + res) ----------------------------------------------------------
        {
            unsigned char* ob = outp + oorg;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                unsigned char* d = ob + (unsigned)(RW(t) * rowb_o) + oo + CG(t) * 16 * OPX;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    f32x4 y = acc[m][t];
                    if constexpr (COUT == 24) {
                        if (m == 1) y = (f32x4){c24_fold1(y[0]), c24_fold1(y[1]), c24_fold1(y[2]), c24_fold1(y[3])};
                    } else {
                        const unsigned eo = (unsigned)(RW(t) * rowb_o) + oo + CG(t) * 16 * OPX;
                        if (p.mul && okt[t]) mv[m][t] = *reinterpret_cast<const f16x4*>(mulp + oorg + eo + 32 * m);
                        if (p.res && okt[t]) rv[m][t] = *reinterpret_cast<const f16x4*>(resp + oorg + eo + 32 * m);
                    }
                    if (p.act_slope != 1.0f) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[i] = fmaxf(y[i], y[i] * p.act_slope);
                    }
                    if (p.mul) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[i] *= (float)mv[m][t][i];
                    }
                    if (p.res) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[i] += (float)rv[m][t][i];
                    }
                    if (p.post_slope != 1.0f) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[i] = fmaxf(y[i], y[i] * p.post_slope);
                    }
                    const f16x4 o = {(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
                    if (okt[t] && (COUT != 24 || m == 0 || q < 2)) *reinterpret_cast<f16x4*>(d + 32 * m) = o;
                }
            }
        }
        if constexpr (CONF == 0) { if (has_next) __syncthreads(); }  // next x tile visible
    }
#undef RW
#undef CG
}

template <int COUT, int NCG0, int NCG1, int NWV, int TH, int WPS, int SHUF = 0, int CONF = 0, int HALF = 0, int MM = 0>
static int launch_c24(C24Args& a, hipStream_t st) {
    constexpr int NZ = HALF ? 2 : SHUF == 0 ? 1 : SHUF == 24 ? 2 : 4;   // row groups of the pixel-shuffle / channel-half variants (blockIdx.y)
    constexpr int NCG = NCG0 + NCG1, PS = NCG | 1;
    constexpr int LDS = c24_steps(NCG) * (COUT == 24 ? 3 : COUT == 3 ? 1 : COUT / 8) * 1024 + (COUT == 48 ? 256 : 128) + (TH + 2) * C24_XW * PS * 16 +
                        (CONF ? 2 * (TH + 4) * (C24_XW + 2) * 4 + 18 * 16 * 4 + 64 : 0);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_done[RV_MAX_DEVICES] = {};
    static int occ_dev[RV_MAX_DEVICES] = {};
    const int dev = rv_device();
    if (!attr_done[dev]) {
        RV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv24_kernel<COUT, NCG0, NCG1, NWV, TH, WPS, SHUF, CONF, HALF, MM>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        int occ = 0;
        RV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv24_kernel<COUT, NCG0, NCG1, NWV, TH, WPS, SHUF, CONF, HALF, MM>, NWV * 64, LDS));
        occ_dev[dev] = occ < 1 ? 1 : occ;
        attr_done[dev] = true;
    }
    a.tiles_x = rv_cdiv(a.w, C24_TW);
    a.tpm = a.tiles_x * rv_cdiv(a.h, TH);
    a.n_tiles = a.tpm * (MM ? a.batch : 1);
    int cap = (rv_stream_cus(st) * occ_dev[dev] / NZ) & ~7;
    if (cap < 8) cap = 8;
    a.grid = a.n_tiles < cap ? a.n_tiles : cap;
    hipLaunchKernelGGL((conv24_kernel<COUT, NCG0, NCG1, NWV, TH, WPS, SHUF, CONF, HALF, MM>), dim3(a.grid, NZ), dim3(NWV * 64), LDS, st, a);
    RV_LAUNCH_CHECK();
    return 0;
}

extern "C" int refvsr_conv24_supported(int c0, int c1) {
    return (c0 == 24 && c1 == 0) || (c0 == 16 && c1 == 0) || (c0 == 8 && c1 == 24) || (c0 == 24 && c1 == 24);
}
extern "C" int refvsr_conv48_supported(int c0, int c1) {
    return (c0 == 48 && c1 == 0) || (c0 == 16 && c1 == 0) || (c0 == 48 && c1 == 48) || (c0 == 8 && c1 == 48);
}
extern "C" int refvsr_conv32_supported(int c0, int c1) { return (c0 == 32 && c1 == 0) || (c0 == 8 && c1 == 0); }

extern "C" int refvsr_conv24_blob_bytes(int c0, int c1) {
    if (!refvsr_conv24_supported(c0, c1)) return -1;
    return c24_steps((c0 + c1) / 8) * 3 * 1024 + 128;
}
extern "C" int refvsr_conv32_blob_bytes(int c0, int c1) {
    if (!refvsr_conv32_supported(c0, c1)) return -1;
    return c24_steps((c0 + c1) / 8) * 4 * 1024 + 128;
}
extern "C" int refvsr_conv48_blob_bytes(int c0, int c1) {
    if (!refvsr_conv48_supported(c0, c1)) return -1;
    if (c0 == 48 && c1 == 48) return 2 * (c24_steps(12) * 3 * 1024 + 128);   // two channel-half blobs of the 24-output layout, back to back
    return c24_steps((c0 + c1) / 8) * 6 * 1024 + 256;
}

extern "C" int refvsr_conv_shuffle2_supported(int c) { return c == 24 || c == 48; }
extern "C" int refvsr_conv_shuffle2_blob_bytes(int c) {
    if (!refvsr_conv_shuffle2_supported(c)) return -1;
    return (c == 24 ? 2 : 4) * (c24_steps(c / 8) * 6 * 1024 + 256);
}

extern "C" int refvsr_conv24_kblock(int ncg, int s, int q) {
    if (c24_steps(ncg) == 0 || s < 0 || s >= c24_steps(ncg) || q < 0 || q > 3) return -2;
    return c24_kblock(ncg, s, q);
}

static int c24_fill(C24Args& a, const char* who, int cout, const void* src0, const void* src1, int c1, int h, int w, const void* blob,
                    float act_slope, const void* mul, const void* res, float post_slope, void* out) {
    RV_CHECK(src0 && out && blob && h > 0 && w > 0, "%s: bad args", who);
    RV_CHECK((c1 == 0) == (src1 == nullptr), "%s: src1 / c1 mismatch", who);
    RV_CHECK(((uintptr_t)blob & 15) == 0, "%s: blob must be 16-byte aligned", who);
    RV_CHECK(act_slope >= 0.f && act_slope <= 1.f && post_slope >= 0.f && post_slope <= 1.f, "%s: activation slopes must lie in [0, 1]", who);
    RV_CHECK(src0 != out && src1 != out, "%s: in-place operation is not supported", who);
    RV_CHECK((long long)h * w * cout * 2 < (1ll << 31), "%s: map too large for 32-bit offsets", who);
    RV_CHECK(refvsr_init() == 0, "init failed");
    memset(&a, 0, sizeof(a));
    a.src0 = (const unsigned char*)src0; a.src1 = (const unsigned char*)src1; a.out = (unsigned char*)out;
    a.blob = (const unsigned char*)blob; a.mul = (const unsigned char*)mul; a.res = (const unsigned char*)res;
    a.h = h; a.w = w; a.act_slope = act_slope; a.post_slope = post_slope;
    return 0;
}

extern "C" int refvsr_conv24(const void* src0, int c0, const void* src1, int c1, int h, int w, const void* blob, float act_slope,
                             const void* mul, const void* res, float post_slope, void* out, void* stream) {
    RV_CHECK(refvsr_conv24_supported(c0, c1), "conv24: %d + %d input channels not supported", c0, c1);
    C24Args a;
    if (c24_fill(a, "conv24", 24, src0, src1, c1, h, w, blob, act_slope, mul, res, post_slope, out)) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (c0 == 24 && c1 == 0) return launch_c24<24, 3, 0, 8, 8, 4>(a, st);
    if (c0 == 16 && c1 == 0) return launch_c24<24, 2, 0, 8, 8, 4>(a, st);
    if (c0 == 8 && c1 == 24) return launch_c24<24, 1, 3, 8, 8, 4>(a, st);
    return launch_c24<24, 3, 3, 8, 8, 4>(a, st);
}

// The same conv over `batch` maps of one geometry in ONE launch (ABI 11): the per-map operands are host arrays of device pointers.
extern "C" int refvsr_conv24_batch(const void* const* src0, int c0, const void* const* src1, int c1, int batch, int h, int w, const void* blob,
                                   float act_slope, const void* const* mul, const void* const* res, float post_slope, void* const* out,
                                   void* stream) {
    RV_CHECK(refvsr_conv24_supported(c0, c1), "conv24_batch: %d + %d input channels not supported", c0, c1);
    RV_CHECK(src0 && out && batch >= 1 && batch <= REFVSR_MAX_MAPS, "conv24_batch: 1..%d maps per launch", REFVSR_MAX_MAPS);
    RV_CHECK((c1 == 0) == (src1 == nullptr), "conv24_batch: src1 / c1 mismatch");
    C24Args a;
    if (c24_fill(a, "conv24_batch", 24, src0[0], src1 ? src1[0] : nullptr, c1, h, w, blob, act_slope, mul ? mul[0] : nullptr,
                 res ? res[0] : nullptr, post_slope, out[0])) return 1;
    a.batch = batch;
    for (int b = 0; b < batch; ++b) {
        RV_CHECK(src0[b] && out[b] && (!src1 || src1[b]) && (!mul || mul[b]) && (!res || res[b]), "conv24_batch: null map pointer (map %d)", b);
        for (int c = 0; c < batch; ++c)
            RV_CHECK(src0[c] != out[b] && (!src1 || src1[c] != out[b]) && (c == b || out[c] != out[b]), "conv24_batch: in-place operation is not supported");
        a.bsrc0[b] = (const unsigned char*)src0[b]; a.bsrc1[b] = src1 ? (const unsigned char*)src1[b] : nullptr;
        a.bout[b] = (unsigned char*)out[b];
        a.bmul[b] = mul ? (const unsigned char*)mul[b] : nullptr; a.bres[b] = res ? (const unsigned char*)res[b] : nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    if (c0 == 24 && c1 == 0) return launch_c24<24, 3, 0, 8, 8, 4, 0, 0, 0, 1>(a, st);
    if (c0 == 16 && c1 == 0) return launch_c24<24, 2, 0, 8, 8, 4, 0, 0, 0, 1>(a, st);
    if (c0 == 8 && c1 == 24) return launch_c24<24, 1, 3, 8, 8, 4, 0, 0, 0, 1>(a, st);
    return launch_c24<24, 3, 3, 8, 8, 4, 0, 0, 0, 1>(a, st);
}

// 32 output channels (AlignedConv2d, RefVSR_/alignment.py:18-24,53-100: the 3 -> 32 stem and the 32 -> 32 convs of its ResBlocks)
extern "C" int refvsr_conv32(const void* src0, int c0, const void* src1, int c1, int h, int w, const void* blob, float act_slope,
                             const void* mul, const void* res, float post_slope, void* out, void* stream) {
    RV_CHECK(refvsr_conv32_supported(c0, c1), "conv32: %d + %d input channels not supported", c0, c1);
    C24Args a;
    if (c24_fill(a, "conv32", 32, src0, src1, c1, h, w, blob, act_slope, mul, res, post_slope, out)) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (c0 == 32) return launch_c24<32, 4, 0, 8, 8, 4>(a, st);
    return launch_c24<32, 1, 0, 8, 8, 4>(a, st);
}

extern "C" int refvsr_conv48(const void* src0, int c0, const void* src1, int c1, int h, int w, const void* blob, float act_slope,
                             const void* mul, const void* res, float post_slope, void* out, void* stream) {
    RV_CHECK(refvsr_conv48_supported(c0, c1), "conv48: %d + %d input channels not supported", c0, c1);
    C24Args a;
    if (c24_fill(a, "conv48", 48, src0, src1, c1, h, w, blob, act_slope, mul, res, post_slope, out)) return 1;
    hipStream_t st = (hipStream_t)stream;
    // 84 KB of weights + 18 x 34-pixel tile: one workgroup per CU.  Sixteen waves with two pixel groups each (default), or eight
    // waves with four (A/B knob REFVSR_CONV48_WAVES=8: 37 % fewer LDS fragment reads, half the waves per SIMD)
    static const bool w8 = getenv("REFVSR_CONV48_WAVES") && atoi(getenv("REFVSR_CONV48_WAVES")) == 8;
    if (c0 == 48 && c1 == 48) return launch_c24<24, 6, 6, 16, 8, 4, 0, 0, 1>(a, st);      // two channel halves on blockIdx.y
    // 8 + 48 -> 48: the input conv of ResidualBlocksWithInputConv on cat([lr, feat]) (RefVSR.py:340-343): NCG = 7 plan, 108 KB of
    // weights resident next to the 38 KB tile, sixteen waves with one pixel group each
    if (c0 == 8 && c1 == 48) return launch_c24<48, 1, 6, 16, 8, 4>(a, st);
    if (c0 == 48) return w8 ? launch_c24<48, 6, 0, 8, 16, 2>(a, st) : launch_c24<48, 6, 0, 16, 16, 4>(a, st);
    return launch_c24<48, 2, 0, 8, 8, 4>(a, st);
}

// act(C -> 4 C 3x3 conv + bias) through F.pixel_shuffle(2) on fp16 HWC maps: src [h][w][C] -> out [2h][2w][C].  blobs: the 2 (C = 24) or
// 4 (C = 48) row-group blobs of refvsr_amd/packing.py:pack_conv_shuffle2, back to back.
extern "C" int refvsr_conv_shuffle2(const void* src, int c, int h, int w, const void* blobs, float act_slope, void* out, void* stream) {
    RV_CHECK(refvsr_conv_shuffle2_supported(c), "conv_shuffle2: %d channels not supported", c);
    C24Args a;
    if (c24_fill(a, "conv_shuffle2", 4 * c, src, nullptr, 0, h, w, blobs, act_slope, nullptr, nullptr, 1.0f, out)) return 1;   // (4 c: the 2h x 2w x c output map)
    hipStream_t st = (hipStream_t)stream;
    if (c == 24) return launch_c24<48, 3, 0, 8, 8, 4, 24>(a, st);
    return launch_c24<48, 6, 0, 16, 16, 4, 48>(a, st);
}

// PixelShufflePack over `batch` maps in ONE launch (ABI 11, C = 24: upsample1 of the RAP steps, RefVSR.py:138)
extern "C" int refvsr_conv_shuffle2_batch(const void* const* src, int batch, int c, int h, int w, const void* blobs, float act_slope,
                                          void* const* out, void* stream) {
    RV_CHECK(c == 24, "conv_shuffle2_batch: %d channels not supported (24)", c);
    RV_CHECK(src && out && batch >= 1 && batch <= REFVSR_MAX_MAPS, "conv_shuffle2_batch: 1..%d maps per launch", REFVSR_MAX_MAPS);
    C24Args a;
    if (c24_fill(a, "conv_shuffle2_batch", 4 * c, src[0], nullptr, 0, h, w, blobs, act_slope, nullptr, nullptr, 1.0f, out[0])) return 1;
    a.batch = batch;
    for (int b = 0; b < batch; ++b) {
        RV_CHECK(src[b] && out[b], "conv_shuffle2_batch: null map pointer (map %d)", b);
        for (int c2 = 0; c2 < batch; ++c2) RV_CHECK(src[c2] != out[b] && (c2 == b || out[c2] != out[b]), "conv_shuffle2_batch: in-place operation is not supported");
        a.bsrc0[b] = (const unsigned char*)src[b]; a.bout[b] = (unsigned char*)out[b];
    }
    return launch_c24<48, 3, 0, 8, 8, 4, 24, 0, 0, 1>(a, (hipStream_t)stream);
}

// The confidence fusions of AA_AF_conf_prop / compute_up in ONE launch (RefVSR.py:47-52 conf_fusion / conf_fusion2 /
// conf_fusion_BWFW, called at :130, :141-142, :107-109):
//   alpha = lrelu_{slope1}( conv3x3_{16 -> cout}( lrelu_{slope0}( conv3x3_{2 -> 16}( P ) ) ) ),
//   P = cat[conf_a, conf_b]  (up = 1)   |   clamp01( F.interpolate(cat[conf_a, conf_b], x2, bicubic) )  (up = 2),
// both convs zero padded.  conf_a / conf_b: planar fp32 [h][w]; w0 / b0: fp32 [16][2][3][3] / [16] (device); blob: the 16 -> cout
// blob of refvsr_conv24 / refvsr_conv48 (cout = 24 | 48); alpha: fp16 HWC [up h][up w][cout]; conf_max (optional, up = 1):
// max(conf_a, conf_b) [h][w] (RefVSR.py:147).  Bit-identical to torch.cat + [refvsr_resize +] refvsr_conv_direct_f32 +
// refvsr_conv24 / 48 [+ refvsr_max2].
extern "C" int refvsr_conf_alpha(const float* conf_a, const float* conf_b, int h, int w, int up, const float* w0, const float* b0,
                                 float slope0, const void* blob, int cout, float slope1, void* alpha, float* conf_max, void* stream) {
    RV_CHECK(conf_a && conf_b && w0 && b0 && alpha, "conf_alpha: null argument");
    RV_CHECK(up == 1 || up == 2, "conf_alpha: up must be 1 or 2");
    RV_CHECK(cout == 24 || cout == 48, "conf_alpha: %d output channels not supported (24 | 48)", cout);
    RV_CHECK(conf_max == nullptr || up == 1, "conf_alpha: the max by-product exists at up = 1 only");
    RV_CHECK(slope0 >= 0.f && slope0 <= 1.f, "conf_alpha: activation slopes must lie in [0, 1]");
    C24Args a;
    if (c24_fill(a, "conf_alpha", cout, conf_a, nullptr, 0, up * h, up * w, blob, slope1, nullptr, nullptr, 1.0f, alpha)) return 1;
    a.src0 = nullptr;
    a.conf_a = conf_a; a.conf_b = conf_b; a.cw0 = w0; a.cb0 = b0; a.conf_max = conf_max; a.ch = h; a.cw = w; a.slope0 = slope0;
    hipStream_t st = (hipStream_t)stream;
    if (cout == 24) return up == 1 ? launch_c24<24, 2, 0, 8, 8, 4, 0, 1>(a, st) : launch_c24<24, 2, 0, 8, 8, 4, 0, 2>(a, st);
    return up == 1 ? launch_c24<48, 2, 0, 8, 8, 4, 0, 1>(a, st) : launch_c24<48, 2, 0, 8, 8, 4, 0, 2>(a, st);
}

// The confidence fusion over `batch` pairs of maps in ONE launch (ABI 11, cout = 24)
extern "C" int refvsr_conf_alpha_batch(const float* const* conf_a, const float* const* conf_b, int batch, int h, int w, int up, const float* w0,
                                       const float* b0, float slope0, const void* blob, int cout, float slope1, void* const* alpha,
                                       float* const* conf_max, void* stream) {
    RV_CHECK(conf_a && conf_b && w0 && b0 && alpha && batch >= 1 && batch <= REFVSR_MAX_MAPS, "conf_alpha_batch: bad args (1..%d maps)", REFVSR_MAX_MAPS);
    RV_CHECK(up == 1 || up == 2, "conf_alpha_batch: up must be 1 or 2");
    RV_CHECK(cout == 24, "conf_alpha_batch: %d output channels not supported (24)", cout);
    RV_CHECK(conf_max == nullptr || up == 1, "conf_alpha_batch: the max by-product exists at up = 1 only");
    RV_CHECK(slope0 >= 0.f && slope0 <= 1.f, "conf_alpha_batch: activation slopes must lie in [0, 1]");
    C24Args a;
    if (c24_fill(a, "conf_alpha_batch", cout, conf_a[0], nullptr, 0, up * h, up * w, blob, slope1, nullptr, nullptr, 1.0f, alpha[0])) return 1;
    a.src0 = nullptr;
    a.conf_a = conf_a[0]; a.conf_b = conf_b[0]; a.cw0 = w0; a.cb0 = b0; a.conf_max = conf_max ? conf_max[0] : nullptr; a.ch = h; a.cw = w; a.slope0 = slope0;
    a.batch = batch;
    for (int b = 0; b < batch; ++b) {
        RV_CHECK(conf_a[b] && conf_b[b] && alpha[b] && (!conf_max || conf_max[b]), "conf_alpha_batch: null map pointer (map %d)", b);
        a.bconf_a[b] = conf_a[b]; a.bconf_b[b] = conf_b[b]; a.bconf_max[b] = conf_max ? conf_max[b] : nullptr;
        a.bout[b] = (unsigned char*)alpha[b];
    }
    hipStream_t st = (hipStream_t)stream;
    return up == 1 ? launch_c24<24, 2, 0, 8, 8, 4, 0, 1, 0, 1>(a, st) : launch_c24<24, 2, 0, 8, 8, 4, 0, 2, 0, 1>(a, st);
}

// The output head in ONE launch (RefVSR.py:92,118,288,297: conv_last 3x3 C -> 3, + F.interpolate(lr_centre, scale, bicubic).clamp(0, 1),
// final clamp): out planar fp32 [3][h][w] = clamp( conv(src) + bias + clamp01(bicubic(base_lr)), 0, 1 ).  src: fp16 HWC [h][w][c],
// c = 24 | 48; base_lr: planar fp32 [3][bh][bw] with h / bh == w / bw the SR factor of this stage; blob: refvsr_conv_last_blob_bytes(c)
// bytes = [S K-steps x ONE fragment x 64 lanes x 8 halfs][32 bias floats]: fragment rows 0-2 = hi(W[r]), rows 8-10 = lo(W[r - 8]),
// K-blocks by refvsr_conv24_kblock(c / 8, s, q) (refvsr_amd/packing.py:pack_conv_last).  Replaces refvsr_resize (the x4 base map)
// + refvsr_conv_mfma's planar mode: the 3-channel base never exists in HBM, one MFMA per K-step instead of two 16-row tiles.
extern "C" int refvsr_conv_last_supported(int c) { return c == 24 || c == 48; }
extern "C" int refvsr_conv_last_blob_bytes(int c) { return refvsr_conv_last_supported(c) ? c24_steps(c / 8) * 1024 + 128 : -1; }
extern "C" int refvsr_conv_last_fmt(const void* src, int c, int h, int w, const void* blob, const float* base_lr, int bh, int bw,
                                    void* out, int out_fmt, void* stream);
extern "C" int refvsr_conv_last(const void* src, int c, int h, int w, const void* blob, const float* base_lr, int bh, int bw,
                                float* out, void* stream) {
    return refvsr_conv_last_fmt(src, c, h, w, blob, base_lr, bh, bw, out, REFVSR_RESULT_F32, stream);
}
extern "C" int refvsr_conv_last_fmt(const void* src, int c, int h, int w, const void* blob, const float* base_lr, int bh, int bw,
                                    void* out, int out_fmt, void* stream) {
    RV_CHECK(out_fmt >= REFVSR_RESULT_F32 && out_fmt <= REFVSR_RESULT_U8, "conv_last: unknown result format %d", out_fmt);
    RV_CHECK(refvsr_conv_last_supported(c), "conv_last: %d input channels not supported (24 | 48)", c);
    RV_CHECK(base_lr && bh > 0 && bw > 0 && h % bh == 0 && w % bw == 0 && h / bh == w / bw, "conv_last: base frame %dx%d does not divide the output %dx%d", bh, bw, h, w);
    C24Args a;
    if (c24_fill(a, "conv_last", c, src, nullptr, 0, h, w, blob, 1.0f, nullptr, nullptr, 1.0f, out)) return 1;
    a.base_lr = base_lr; a.bh = bh; a.bw = bw; a.base_step = (float)bh / (float)h; a.out_fmt = out_fmt;
    hipStream_t st = (hipStream_t)stream;
    if (c == 24) return launch_c24<3, 3, 0, 8, 8, 4>(a, st);
    return launch_c24<3, 6, 0, 8, 8, 4>(a, st);
}
