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
// LDS: [fragments: S K-steps x 3 x 1 KiB][bias: 32 floats][x tile: 10 x 34 pixels x PS slots of 16 bytes, PS = NCG | 1 (odd
// pixel stride: bank-conflict-free B reads with the pixel permutation of common.h)].  K plans (c24_kblock): the K-blocks of
// one window row are the slots u = tx * PS + cg; a K-step takes four of them whose offsets are (step immediate) + (one of
// <= 4 per-lane patterns), the two blocks a ds_read_b128 lane group mixes having slot offsets of equal parity:
//   NCG 3: rows of 9 consecutive slots -> 2 steps of {0,2,1,3} + 4a per row, then u = 8 of the three rows   (7 steps)
//   NCG 4: one step per tap, cg = {0,2,1,3}                                                                  (9 steps)
//   NCG 2: u = {0,4,1,3} per row, then u = 6 | 7 of rows 0,1, then of row 2                                  (5 steps)
//   NCG 6: one step per tap with cg = {0,2,1,3}; cg 4,5 of taps (tx 0, tx 1) per row; of tx 2 for rows 0,1; for row 2   (14 steps)
#include <type_traits>
#include <utility>

#include "common.h"

namespace {
constexpr int C24_TH = 8, C24_TW = 32, C24_XH = 10, C24_XW = 34, C24_NPX = C24_XH * C24_XW;   // 340 staged pixels

__host__ __device__ constexpr int c24_steps(int ncg) { return ncg == 2 ? 5 : ncg == 3 ? 7 : ncg == 4 ? 9 : ncg == 6 ? 14 : 0; }

// K-block (K-step s, quarter q) -> ty << 16 | tx << 8 | cg, or -1 for a zero block
__host__ __device__ constexpr int c24_kblock(int ncg, int s, int q) {
    const int perm[4] = {0, 2, 1, 3};
    int ty = 0, tx = 0, cg = 0;
    if (ncg == 3) {
        if (s < 6) { const int u = 4 * (s & 1) + perm[q]; ty = s >> 1; tx = u / 3; cg = u % 3; }
        else { if (q == 3) return -1; ty = q; tx = 2; cg = 2; }
    } else if (ncg == 4) {
        ty = s / 3; tx = s % 3; cg = perm[q];
    } else if (ncg == 2) {
        if (s < 3) { const int u4[4] = {0, 4, 1, 3}; ty = s; tx = u4[q] / 3; cg = u4[q] % 3; }
        else if (s == 3) { ty = q & 1; tx = 2; cg = q >> 1; }
        else { if (q & 1) return -1; ty = 2; tx = 2; cg = q >> 1; }
    } else if (ncg == 6) {
        if (s < 9) { ty = s / 3; tx = s % 3; cg = perm[q]; }
        else if (s < 12) { const int txs[4] = {0, 1, 0, 1}, cgs[4] = {4, 5, 5, 4}; ty = s - 9; tx = txs[q]; cg = cgs[q]; }
        else if (s == 12) { ty = q & 1; tx = 2; cg = 4 + (q >> 1); }
        else { if (q & 1) return -1; ty = 2; tx = 2; cg = 4 + (q >> 1); }
    } else {
        return -2;
    }
    return (ty << 16) | (tx << 8) | cg;
}

// LDS byte offset of K-block (s, q) relative to a window origin (zero blocks read their left neighbour's address)
__host__ __device__ constexpr int c24_off(int ncg, int s, int q) {
    int kb = c24_kblock(ncg, s, q);
    if (kb < 0) kb = c24_kblock(ncg, s, q - 1);
    const int ps = ncg | 1;
    return (kb >> 16) * (C24_XW * ps * 16) + (((kb >> 8) & 255) * ps + (kb & 255)) * 16;
}
// pattern of a K-step: steps of one pattern differ only by an immediate
__host__ __device__ constexpr int c24_pat(int ncg, int s) {
    return ncg == 3 ? (s < 6 ? 0 : 1) : ncg == 4 ? 0 : ncg == 2 ? (s < 3 ? 0 : s - 2) : (s < 9 ? 0 : s < 12 ? 1 : s - 10);
}
__host__ __device__ constexpr int c24_npat(int ncg) { return ncg == 3 ? 2 : ncg == 4 ? 1 : ncg == 2 ? 3 : 4; }
// first K-step of a pattern
__host__ __device__ constexpr int c24_pat_step(int ncg, int p) {
    return ncg == 3 ? (p ? 6 : 0) : ncg == 4 ? 0 : ncg == 2 ? (p ? p + 2 : 0) : (p == 0 ? 0 : p == 1 ? 9 : p + 10);
}

// compile-time proof of the plans: every K-block of the 3 x 3 x ncg window exactly once, K-steps of one pattern differ by an
// immediate only, the quarters (0, 1) and (2, 3) of a step read slots of equal parity (bank-conflict-free ds_read_b128)
__host__ __device__ constexpr bool c24_plan_ok(int ncg) {
    const int S = c24_steps(ncg);
    int seen[3 * 3 * 8] = {};
    for (int s = 0; s < S; ++s) {
        const int ps = c24_pat_step(ncg, c24_pat(ncg, s));
        if (c24_pat(ncg, ps) != c24_pat(ncg, s)) return false;
        for (int q = 0; q < 4; ++q) {
            const int kb = c24_kblock(ncg, s, q);
            if (kb >= 0) {
                const int ty = kb >> 16, tx = (kb >> 8) & 255, cg = kb & 255;
                if (ty > 2 || tx > 2 || cg >= ncg) return false;
                seen[(ty * 3 + tx) * 8 + cg] += 1;
            } else if (q == 0) {
                return false;
            }
            if (c24_off(ncg, s, q) - c24_off(ncg, s, 0) != c24_off(ncg, ps, q) - c24_off(ncg, ps, 0)) return false;
        }
        if (((c24_off(ncg, s, 0) ^ c24_off(ncg, s, 1)) & 16) || ((c24_off(ncg, s, 2) ^ c24_off(ncg, s, 3)) & 16)) return false;
    }
    for (int t = 0; t < 9; ++t)
        for (int cg = 0; cg < ncg; ++cg)
            if (seen[t * 8 + cg] != 1) return false;
    for (int p = 0; p < c24_npat(ncg); ++p)
        if (c24_pat(ncg, c24_pat_step(ncg, p)) != p) return false;
    return true;
}
static_assert(c24_plan_ok(2) && c24_plan_ok(3) && c24_plan_ok(4) && c24_plan_ok(6), "conv24 K plan");

template <class F, int... I>
__device__ __forceinline__ void c24_static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
}  // namespace

struct C24Args {
    const unsigned char* src0; const unsigned char* src1; unsigned char* out;
    const unsigned char* blob; const unsigned char* mul; const unsigned char* res;
    int h, w, tiles_x, n_tiles, grid;
    float act_slope, post_slope;
};

__device__ __forceinline__ float c24_fold1(const float a) {       // lane l: a[l] + a[l ^ 32] (see resblock24.hip:rb_fold1)
    const unsigned u = __float_as_uint(a);
    const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned x0 = pr[0], x1 = pr[1];
    return __uint_as_float(x0) + __uint_as_float(x1);
}

template <int NCG0, int NCG1, int NWV>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(NWV / 2, NWV / 2))) void conv24_kernel(C24Args p) {
    constexpr int NCG = NCG0 + NCG1, PS = NCG | 1, PXB = PS * 16, ROWB = C24_XW * PXB;
    constexpr int S = c24_steps(NCG), NPAT = c24_npat(NCG);
    constexpr int WB = S * 3 * 1024, BIAS = WB, XT = WB + 128;
    constexpr int NT = NWV * 64, T = 16 / NWV;
    constexpr int NCH = C24_NPX * NCG, KCH = (NCH + NT - 1) / NT;
    constexpr int PIXB0 = NCG0 * 16, PIXB1 = NCG1 * 16;
    constexpr int OPX = 48;                                         // bytes per pixel of the 24-channel out / mul / res maps
    static_assert(S > 0 && T >= 1, "unsupported shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    asm volatile("" :: "s"(p.src0), "s"(p.src1), "s"(p.out), "s"(p.blob), "s"(p.mul), "s"(p.res), "s"(p.h), "s"(p.w), "s"(p.tiles_x),
                 "s"(p.n_tiles), "s"(p.grid), "s"(p.act_slope), "s"(p.post_slope));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weights + bias: global -> LDS (1 KiB per wave instruction), issued first
    {
        constexpr int NPC = WB / 1024;                               // full pieces; the 128-byte bias tail: 8 lanes
        const unsigned char* g = p.blob + lane * 16;
#pragma unroll
        for (int j = 0; j < (NPC + NWV - 1) / NWV; ++j) {
            const int c = wave + j * NWV;
            if (c < NPC)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + c * 1024),
                                                 (__attribute__((address_space(3))) void*)(smem + c * 1024), 16, 0, 0);
        }
        if (wave == NPC % NWV && lane < 8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + NPC * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + NPC * 1024), 16, 0, 0);
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
    auto x_fetch = [&](const int t) {
        const int tyi = t / p.tiles_x;
        const int ty0 = tyi * C24_TH, tx0 = (t - tyi * p.tiles_x) * C24_TW;
        const bool interior = ty0 >= 1 && ty0 + C24_TH + 1 <= p.h && tx0 >= 1 && tx0 + C24_TW + 1 <= p.w;
        const long long org = (long long)(ty0 - 1) * p.w + (tx0 - 1);          // origin pixel (may lie outside the frame)
        const unsigned char* b0 = p.src0 + org * PIXB0;
        const unsigned char* b1 = NCG1 ? p.src1 + org * PIXB1 : b0;
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
                const unsigned char* g = s1 ? p.src1 + pix * PIXB1 + (cg - NCG0) * 16 : p.src0 + pix * PIXB0 + cg * 16;
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
    int tl, k_hi;
    rv_tile_range(p.n_tiles, p.grid, tl, k_hi);
    if (tl < k_hi) x_fetch(tl);
    __builtin_amdgcn_sched_barrier(0);                               // weights and first tile in flight before the rest of the set-up

    // ---- per-lane constants
    const int q = lane >> 4;
    const int lp = rv_pix16(lane & 15);
    const int la = lane * 16;
    const int oy0 = (wave * T) >> 1;
    auto sel4 = [&](const int v0, const int v1, const int v2, const int v3) { return q == 0 ? v0 : q == 1 ? v1 : q == 2 ? v2 : v3; };
    int pb[T];                                                       // window origin of each output group + this lane's pattern-0 offset
#pragma unroll
    for (int t = 0; t < T; ++t)
        pb[t] = XT + (oy0 + (t >> 1)) * ROWB + ((t & 1) * 16 + lp) * PXB +
                sel4(c24_off(NCG, 0, 0), c24_off(NCG, 0, 1), c24_off(NCG, 0, 2), c24_off(NCG, 0, 3)) - c24_off(NCG, 0, 0);
    int pd[NPAT];                                                    // pattern p relative to pattern 0, per lane
#pragma unroll
    for (int pp = 0; pp < NPAT; ++pp) {
        const int s = c24_pat_step(NCG, pp);
        pd[pp] = (sel4(c24_off(NCG, s, 0), c24_off(NCG, s, 1), c24_off(NCG, s, 2), c24_off(NCG, s, 3)) - c24_off(NCG, s, 0)) -
                 (sel4(c24_off(NCG, 0, 0), c24_off(NCG, 0, 1), c24_off(NCG, 0, 2), c24_off(NCG, 0, 3)) - c24_off(NCG, 0, 0));
    }
    const int rowb_o = p.w * OPX;
    const unsigned oo = (unsigned)(oy0 * rowb_o + lp * OPX + q * 8);  // this lane's channels 4q.. of group 0, relative to the tile origin

    if (tl < k_hi) x_park();
    __syncthreads();                                                 // weights, bias, first tile

    for (; tl < k_hi; ++tl) {
        const bool has_next = tl + 1 < k_hi;
        const int tyi = tl / p.tiles_x;
        const int ty0 = tyi * C24_TH, tx0 = (tl - tyi * p.tiles_x) * C24_TW;
        const bool interior = ty0 >= 1 && ty0 + C24_TH + 1 <= p.h && tx0 >= 1 && tx0 + C24_TW + 1 <= p.w;
        const long long oorg = ((long long)ty0 * p.w + tx0) * OPX;
        // epilogue operands of this tile, in flight during the K loop
        f16x4 m0[T], m1[T], r0[T], r1[T];
        bool okt[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            okt[t] = true;
            if (!interior) {
                int lpe = lp;
                asm volatile("" : "+v"(lpe));
                okt[t] = ty0 + oy0 + (t >> 1) < p.h && tx0 + (t & 1) * 16 + lpe < p.w;
            }
            const unsigned eo = (unsigned)((t >> 1) * rowb_o) + oo + (t & 1) * 16 * OPX;
            m0[t] = m1[t] = r0[t] = r1[t] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
            if (p.mul && okt[t]) {
                m0[t] = *reinterpret_cast<const f16x4*>(p.mul + oorg + eo);
                if (q < 2) m1[t] = *reinterpret_cast<const f16x4*>(p.mul + oorg + eo + 32);
            }
            if (p.res && okt[t]) {
                r0[t] = *reinterpret_cast<const f16x4*>(p.res + oorg + eo);
                if (q < 2) r1[t] = *reinterpret_cast<const f16x4*>(p.res + oorg + eo + 32);
            }
        }
        if (has_next) x_fetch(tl + 1);                               // next tile: in flight during the K loop

        // ---------------- K loop: acc = bias + conv(x) on the 8 x 32 tile ---------------------------------------------------
        f32x4 a0[T], a1[T];
        {
            const f32x4 bv0 = *reinterpret_cast<const f32x4*>(smem + BIAS + q * 16);          // channels 4q ..
            const f32x4 bv1 = *reinterpret_cast<const f32x4*>(smem + BIAS + 64 + q * 16);     // channels 16 + 4q .. (24..31: zeros)
#pragma unroll
            for (int t = 0; t < T; ++t) { a0[t] = bv0; a1[t] = bv1; }
        }
        {
            uint4 fa[2][3], fb[2][T];
            auto load = [&](auto sc, uint4 (&af)[3], uint4 (&bf)[T]) {
                constexpr int s = decltype(sc)::value;
                constexpr int pp = c24_pat(NCG, s);
                constexpr int imm = c24_off(NCG, s, 0) - c24_off(NCG, c24_pat_step(NCG, pp), 0) + c24_off(NCG, c24_pat_step(NCG, pp), 0);
#pragma unroll
                for (int f = 0; f < 3; ++f) af[f] = *reinterpret_cast<const uint4*>(smem + (s * 3 + f) * 1024 + la);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if constexpr (pp == 0) bf[t] = *reinterpret_cast<const uint4*>(smem + pb[t] + imm);
                    else bf[t] = *reinterpret_cast<const uint4*>(smem + pb[t] + pd[pp] + imm);
                }
            };
            auto mfma = [&](const uint4 (&af)[3], const uint4 (&bf)[T]) {
                const f16x8 a_hi = *reinterpret_cast<const f16x8*>(&af[0]);
                const f16x8 a_lo = *reinterpret_cast<const f16x8*>(&af[1]);
                const f16x8 a_mx = *reinterpret_cast<const f16x8*>(&af[2]);
#pragma unroll
                for (int t = 0; t < T; ++t) a0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, *reinterpret_cast<const f16x8*>(&bf[t]), a0[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < T; ++t) a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_mx, *reinterpret_cast<const f16x8*>(&bf[t]), a1[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < T; ++t) a0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, *reinterpret_cast<const f16x8*>(&bf[t]), a0[t], 0, 0, 0);
            };
            load(std::integral_constant<int, 0>{}, fa[0], fb[0]);
            c24_static_for([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + 1 < S) load(std::integral_constant<int, s + 1>{}, fa[(s + 1) & 1], fb[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                mfma(fa[s & 1], fb[s & 1]);
            }, std::make_integer_sequence<int, S>{});
        }
        if (has_next) {
            __syncthreads();                                         // every wave is done reading the x tile
            x_park();
        }
        // ---------------- epilogue: out = post(act(acc) * mul + res) ----------------------------------------------------------
        {
            unsigned char* ob = p.out + oorg;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x4 y0 = a0[t];
                f32x4 y1 = {c24_fold1(a1[t][0]), c24_fold1(a1[t][1]), c24_fold1(a1[t][2]), c24_fold1(a1[t][3])};
                if (p.act_slope != 1.0f) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { y0[i] = fmaxf(y0[i], y0[i] * p.act_slope); y1[i] = fmaxf(y1[i], y1[i] * p.act_slope); }
                }
                if (p.mul) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { y0[i] *= (float)m0[t][i]; y1[i] *= (float)m1[t][i]; }
                }
                if (p.res) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { y0[i] += (float)r0[t][i]; y1[i] += (float)r1[t][i]; }
                }
                if (p.post_slope != 1.0f) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { y0[i] = fmaxf(y0[i], y0[i] * p.post_slope); y1[i] = fmaxf(y1[i], y1[i] * p.post_slope); }
                }
                const f16x4 o0 = {(f16)y0[0], (f16)y0[1], (f16)y0[2], (f16)y0[3]};
                const f16x4 o1 = {(f16)y1[0], (f16)y1[1], (f16)y1[2], (f16)y1[3]};
                unsigned char* d = ob + (unsigned)((t >> 1) * rowb_o) + oo + (t & 1) * 16 * OPX;
                if (okt[t]) {
                    *reinterpret_cast<f16x4*>(d) = o0;
                    if (q < 2) *reinterpret_cast<f16x4*>(d + 32) = o1;
                }
            }
        }
        if (has_next) __syncthreads();                               // next x tile visible
    }
}

template <int NCG0, int NCG1>
static int launch_c24(C24Args& a, hipStream_t st) {
    constexpr int NCG = NCG0 + NCG1, PS = NCG | 1;
    constexpr int LDS = c24_steps(NCG) * 3 * 1024 + 128 + C24_NPX * PS * 16;
    static bool attr_done[RV_MAX_DEVICES] = {};
    static int occ_dev[RV_MAX_DEVICES] = {};
    const int dev = rv_device();
    if (!attr_done[dev]) {
        RV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv24_kernel<NCG0, NCG1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        int occ = 0;
        RV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv24_kernel<NCG0, NCG1, 8>, 512, LDS));
        occ_dev[dev] = occ < 1 ? 1 : occ;
        attr_done[dev] = true;
    }
    int cap = (rv_num_cus() * occ_dev[dev]) & ~7;
    if (cap < 8) cap = 8;
    a.grid = a.n_tiles < cap ? a.n_tiles : cap;
    hipLaunchKernelGGL((conv24_kernel<NCG0, NCG1, 8>), dim3(a.grid), dim3(512), LDS, st, a);
    RV_LAUNCH_CHECK();
    return 0;
}

extern "C" int refvsr_conv24_supported(int c0, int c1) {
    return (c0 == 24 && c1 == 0) || (c0 == 16 && c1 == 0) || (c0 == 8 && c1 == 24) || (c0 == 24 && c1 == 24);
}

extern "C" int refvsr_conv24_blob_bytes(int c0, int c1) {
    if (!refvsr_conv24_supported(c0, c1)) return -1;
    return c24_steps((c0 + c1) / 8) * 3 * 1024 + 128;
}

extern "C" int refvsr_conv24_kblock(int ncg, int s, int q) {
    if (c24_steps(ncg) == 0 || s < 0 || s >= c24_steps(ncg) || q < 0 || q > 3) return -2;
    return c24_kblock(ncg, s, q);
}

extern "C" int refvsr_conv24(const void* src0, int c0, const void* src1, int c1, int h, int w, const void* blob, float act_slope,
                             const void* mul, const void* res, float post_slope, void* out, void* stream) {
    RV_CHECK(src0 && out && blob && h > 0 && w > 0, "conv24: bad args");
    RV_CHECK(refvsr_conv24_supported(c0, c1), "conv24: %d + %d input channels not supported", c0, c1);
    RV_CHECK((c1 == 0) == (src1 == nullptr), "conv24: src1 / c1 mismatch");
    RV_CHECK(((uintptr_t)blob & 15) == 0, "conv24: blob must be 16-byte aligned");
    RV_CHECK(act_slope >= 0.f && act_slope <= 1.f && post_slope >= 0.f && post_slope <= 1.f, "conv24: activation slopes must lie in [0, 1]");
    RV_CHECK(src0 != out && src1 != out, "conv24: in-place operation is not supported");
    RV_CHECK((long long)h * w * 48 < (1ll << 31), "conv24: map too large for 32-bit offsets");
    RV_CHECK(refvsr_init() == 0, "init failed");
    C24Args a;
    memset(&a, 0, sizeof(a));
    a.src0 = (const unsigned char*)src0; a.src1 = (const unsigned char*)src1; a.out = (unsigned char*)out;
    a.blob = (const unsigned char*)blob; a.mul = (const unsigned char*)mul; a.res = (const unsigned char*)res;
    a.h = h; a.w = w; a.act_slope = act_slope; a.post_slope = post_slope;
    a.tiles_x = rv_cdiv(w, C24_TW);
    a.n_tiles = a.tiles_x * rv_cdiv(h, C24_TH);
    hipStream_t st = (hipStream_t)stream;
    if (c0 == 24 && c1 == 0) return launch_c24<3, 0>(a, st);
    if (c0 == 16 && c1 == 0) return launch_c24<2, 0>(a, st);
    if (c0 == 8 && c1 == 24) return launch_c24<1, 3>(a, st);
    return launch_c24<3, 3>(a, st);
}
