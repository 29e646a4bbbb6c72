// Fused residual block, small-footprint version:  out = post( x + conv2( act( conv1(x) ) ) ),  3x3, C -> C.
// Same arithmetic as resblock_mfma.hip / two refvsr_conv_mfma launches (bit-identical: same K order, same fp16
// rounding points), different shape of the workgroup.  Why: the s_memtime probe of the 8-wave, 16x32-tile kernel
// (tools/probe_resblock.py, profiles/r02_probe_resblock.txt) shows its two K loops running at the matrix-pipe rate
// (8.5k of 17.4k cycles per tile) and the OTHER half of the time going to VALU-only stretches -- prologue, the two
// epilogues (350 / 280 VALU instructions per wave at 2 waves per SIMD ~ 4k / 3k cycles) and barrier skew -- during which
// the matrix pipe idles, because one 154 KB-LDS workgroup per CU runs all its waves in lockstep.  Here:
//
//   * workgroup = 4 waves, output tile 8 x 32, LDS = both weight sets + ONE activation tile (the intermediate map t
//     overwrites the x tile once every wave has finished reading it; the residual x values wait in registers):
//     77 KB for C = 24  ->  two workgroups per CU that drift apart, so one workgroup's epilogue / staging runs under the
//     other's MFMAs (and under workgroups of other streams' launches);
//   * leaner epilogues (activation as max(y, slope*y), border mask applied to the packed halves, chunk bookkeeping of
//     the tile staging computed once per thread);
//   * persistent over tiles in XCD bands, next x tile prefetched into registers during conv2.
//
//   stage   x tile (12 x 36 px, zero padded), w1, w2                global -> LDS
//   phase 1 acc1 = conv1(x) on the 10 x 34 halo region (340 px = 22 sixteen-pixel MFMA tiles, flattened, 6 per wave)
//           residual x values of this lane's outputs -> registers;  barrier;  t = act(acc1 + b1) (0 outside the
//           frame = conv2's zero padding) -> LDS over the x tile;  barrier
//   phase 2 out = post(x + conv2(t) + b2) on the 8 x 32 tile -> global (8-byte HWC channel vectors)
#include <type_traits>

#include "common.h"

#define RL_TH 8
#define RL_TW 32
#define RL_XH (RL_TH + 4)
#define RL_XW (RL_TW + 4)
#define RL_IH (RL_TH + 2)
#define RL_IW (RL_TW + 2)
#define RL_NI (RL_IH * RL_IW)             // 340 intermediate pixels
#define RL_T1 ((RL_NI + 15) / 16)         // 22 phase-1 tiles
// tiles per wave are template constants of the kernel: NWV = 4 waves -> 6 / 4, NWV = 8 waves -> 3 / 2

struct ResLeanArgs {
    const f16* src; f16* out;
    int c, ncg, ps, h, w;
    int G, S;
    float inv_ncg;
    const uint4* w1; const float* b1;
    const uint4* w2; const float* b2;
    float act_slope, post_slope;
    int tab_bytes, w_bytes;               // LDS carve
    int tiles_x, n_tiles;
    int grid;                             // gridDim.x
    unsigned long long* probe;           // debug: per-workgroup s_memtime stamps (refvsr_set_probe), normally null
    int probe_iter;                      // which tile iteration of the workgroup is stamped
};

// Software-pipelined K loop (two fragment sets, unrolled by two) -- the same walk as conv_mfma.hip / resblock_mfma.hip.
template <int MT, int T>
__device__ __forceinline__ void rl_kloop(f32x4 (&acc)[MT][T], const unsigned char* wl, const int* tq,
                                         const unsigned char* src, const int (&pb)[T], const int S, const int lane) {
    constexpr int NA = MT * 2;
    auto load_frag = [&](const int s, const int toff, uint4 (&a)[NA], uint4 (&b)[T]) {
#pragma unroll
        for (int i = 0; i < NA; ++i) a[i] = *reinterpret_cast<const uint4*>(wl + ((size_t)(s * NA + i) * 64 + lane) * 16);
#pragma unroll
        for (int t = 0; t < T; ++t) b[t] = *reinterpret_cast<const uint4*>(src + pb[t] + toff);
    };
    auto mfma_step = [&](const uint4 (&a)[NA], const uint4 (&b)[T]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f16x8 bv = *reinterpret_cast<const f16x8*>(&b[t]);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f16x8 av = *reinterpret_cast<const f16x8*>(&a[m * 2 + h]);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[m][t], 0, 0, 0);
                }
            }
    };
    const int last = S - 1;
    uint4 a0[NA], b0[T], a1[NA], b1[T];
    load_frag(0, tq[0], a0, b0);
    int t1 = tq[min(1, last) * 4];
    for (int s = 0; s < S; s += 2) {
        const int s1 = min(s + 1, last), s2 = min(s + 2, last), s3 = min(s + 3, last);
        const int t2 = tq[s2 * 4];
        load_frag(s1, t1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(a0, b0);
        const int t3 = tq[s3 * 4];
        load_frag(s2, t2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < S) mfma_step(a1, b1);
        t1 = t3;
    }
}

// NWV = waves per workgroup (4 or 8).  WIDE = false: a staged row has 128 chunk slots (up to 3 channel groups),
// true: 256 (up to 7).  With 8 waves each wave holds half the tiles (3 + 2 instead of 6 + 4): <= 128 VGPRs, so the two
// workgroups of a CU put FOUR waves on every SIMD -- the VALU-only stretches (epilogues, staging) issue at twice the
// rate of the 2-waves-per-SIMD shapes, which is what bounds them (tools/probe_resblock.py).
template <int MT, bool WIDE, int NWV>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(NWV / 2, NWV / 2))) void resblock_lean_kernel(ResLeanArgs p) {
    constexpr int NT = NWV * 64;                                  // threads
    constexpr int RL_T1W = (RL_T1 + NWV - 1) / NWV;               // phase-1 tiles per wave
    constexpr int RL_T2W = RL_TH * 2 / NWV;                       // phase-2 tiles per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* tab1 = reinterpret_cast<int*>(smem);                    // K-slot -> byte offset in the x tile (pitch RL_XW)
    int* tab2 = tab1 + p.S * 4;                                   // K-slot -> byte offset in the t tile (pitch RL_IW)
    unsigned char* wl1 = smem + p.tab_bytes;
    unsigned char* wl2 = wl1 + p.w_bytes;
    unsigned char* xt = wl2 + p.w_bytes;                          // x tile [RL_XH][RL_XW][ps*16], then t [RL_IH][RL_IW][ps*16]

    constexpr int CPR = WIDE ? 256 : 128;                         // chunk slots per staged row
    constexpr int RPP = NT / CPR;                                 // rows per staging pass
    constexpr int XP = RL_XH / RPP;                               // staging passes of the x tile
    static_assert(XP * RPP == RL_XH, "staging passes must cover the x tile");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = lane >> 4;
    const int lp = rv_pix16(lane & 15);       // pixel (of a 16-pixel MFMA tile) held by this lane's column
    const int psb = p.ps * 16;
    // every kernel argument the prologue needs, fetched by ONE batch of scalar loads: hipcc otherwise loads the fields of the
    // by-value argument struct where they are first used -- six dependent s_load / s_waitcnt round trips before the first
    // x-tile load could leave (s_memtime probe + ISA)
    asm volatile("" :: "s"(p.src), "s"(p.out), "s"(p.w1), "s"(p.w2), "s"(p.b1), "s"(p.b2), "s"(p.c), "s"(p.ncg), "s"(p.ps),
                 "s"(p.h), "s"(p.w), "s"(p.G), "s"(p.S), "s"(p.inv_ncg), "s"(p.tab_bytes), "s"(p.w_bytes), "s"(p.tiles_x),
                 "s"(p.n_tiles), "s"(p.grid), "s"(p.probe), "s"(p.probe_iter));
#define RL_STAMP(i) do { if (p.probe && tid == 0) p.probe[blockIdx.x * 12 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    RL_STAMP(0);
    if (p.probe && tid == 0) p.probe[blockIdx.x * 12 + 8] = __builtin_amdgcn_s_memrealtime();

    // Start-up: both weight sets go global -> LDS with global_load_lds_dwordx4 (1 KiB per wave instruction, no staging
    // registers, no ds_write pass: the LDS image of a packed weight set IS its global image), issued first; the x tile and the
    // biases follow through registers (they need masking / re-layout).  Everything lands behind ONE barrier; the weights then
    // stay resident for the workgroup's whole life.  An s_memtime probe showed the prologue to be instruction-bound (~300
    // instructions per wave x 16 waves per CU), not latency-bound: the register-staged copy alone was ~80 of them.
    {
        const int nch = p.S * MT * 2;                             // 1 KiB chunks per weight set
        for (int c = wave; c < 2 * nch; c += NWV) {
            const bool second = c >= nch;
            const int cc = second ? c - nch : c;
            const unsigned char* gsrc = reinterpret_cast<const unsigned char*>(second ? p.w2 : p.w1) + cc * 1024 + lane * 16;
            unsigned char* ldst = (second ? wl2 : wl1) + cc * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                             (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
        }
    }

    for (int g = tid; g < p.S * 4; g += NT) {                     // K order: common.h:rv_kslot
        int o1, o2, slot;
        if (g < p.G) {
            const int tap = (int)(((float)g + 0.5f) * p.inv_ncg);
            const int cg = g - tap * p.ncg;
            const int ty = tap / 3;
            const int tx = tap - ty * 3;
            o1 = ((ty * RL_XW + tx) * p.ps + cg) * 16;
            o2 = ((ty * RL_IW + tx) * p.ps + cg) * 16;
            slot = rv_kslot(ty, tx, cg, 3, p.ncg);
        } else {                                                  // zero-weight blocks: offset of the partner's parity
            slot = rv_kpad_slot(g - p.G, 3, p.ncg);
            o1 = o2 = (slot < p.G) ? 0 : (p.ncg > 1 ? 16 : p.ps * 16);
        }
        tab1[slot] = o1;
        tab2[slot] = o2;
    }
    // x-tile staging slot of this thread: row (xrow0 + k * RPP), column xcol, channel group xcg -- computed once
    const int xi = tid & (CPR - 1);
    const int xrow0 = tid / CPR;
    const int xcol = (int)(((float)xi + 0.5f) * p.inv_ncg);
    const int xcg = xi - xcol * p.ncg;
    const bool xslot = xi < RL_XW * p.ncg;
    const int xlds = (xrow0 * RL_XW + xcol) * psb + xcg * 16;     // + k * RPP * RL_XW * psb
    uint4 xv[XP];
    unsigned xmask = 0;                                           // bit k: chunk k lies inside the frame
    // Loads are UNCONDITIONAL (clamped address) and masked when parked: a load under a lane condition makes hipcc wait
    // for it at the end of the conditional block, which would split the prologue into two memory round trips.
    auto x_fetch = [&](const int t) {                             // global -> registers
        const int tyi = t / p.tiles_x;
        const int iy0 = tyi * RL_TH - 2 + xrow0;
        const int ix = (t - tyi * p.tiles_x) * RL_TW - 2 + xcol;
        const bool colok = xslot && ix >= 0 && ix < p.w;
        const f16* base = p.src + (size_t)min(max(ix, 0), p.w - 1) * p.c + xcg * 8;
        xmask = 0;
#pragma unroll
        for (int k = 0; k < XP; ++k) {
            const int iy = iy0 + k * RPP;
            xmask |= (colok && iy >= 0 && iy < p.h) ? (1u << k) : 0u;
            xv[k] = *reinterpret_cast<const uint4*>(base + (size_t)min(max(iy, 0), p.h - 1) * p.w * p.c);
        }
    };
    auto x_park = [&]() {                                         // registers -> LDS (zero outside the frame)
        if (xslot) {
#pragma unroll
            for (int k = 0; k < XP; ++k) {
                const unsigned keep = (xmask >> k) & 1u ? 0xffffffffu : 0u;
                uint4 v = xv[k];
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                *reinterpret_cast<uint4*>(xt + xlds + k * RPP * RL_XW * psb) = v;
            }
        }
    };

    int tl, k_hi;                                                 // this workgroup's tiles (common.h:rv_tile_range)
    rv_tile_range(p.n_tiles, p.grid, tl, k_hi);
    constexpr int k_step = 1;
    if (tl < k_hi) x_fetch(tl);
    // biases: 2 x 32 floats parked in LDS behind the tables (in registers they cost 16 VGPRs for the kernel's whole life)
    float* bl = reinterpret_cast<float*>(tab2 + p.S * 4);
    // (unconditional clamped loads + select: a load under a lane condition is waited for on the spot -- together with
    //  everything else in flight, i.e. the weight and x-tile loads above)
    const int bi_ = min(tid & 31, p.c - 1);
    const float b1v_ = p.b1[bi_], b2v_ = p.b2[bi_];
    const float bias_v = (tid & 31) < p.c ? ((tid & 32) ? b2v_ : b1v_) : 0.f;
    RL_STAMP(1);
    asm volatile("" ::: "memory");                  // loads above, LDS stores below
    if (tid < 64) bl[tid] = bias_v;
    if (tl < k_hi) x_park();
    // per-lane pixel bookkeeping, independent of the tile origin
    int pb1[RL_T1W], pb2[RL_T2W];
#pragma unroll
    for (int t = 0; t < RL_T1W; ++t) {
        const int pix = min((wave * RL_T1W + t) * 16 + lp, RL_NI - 1);
        const int r = (int)(((float)pix + 0.5f) * (1.0f / (float)RL_IW));
        pb1[t] = (r * RL_XW + (pix - r * RL_IW)) * psb;
    }
#pragma unroll
    for (int t = 0; t < RL_T2W; ++t) {
        const int ti = wave * RL_T2W + t;
        pb2[t] = ((ti >> 1) * RL_IW + (ti & 1) * 16 + lp) * psb;
    }
    __syncthreads();                                // tables, weights, first tile
    RL_STAMP(2);

    for (int iter = 0; tl < k_hi; tl += k_step, ++iter) {
        const bool has_next = tl + k_step < k_hi;
        const int tyi = tl / p.tiles_x;
        const int ty0 = tyi * RL_TH, tx0 = (tl - tyi * p.tiles_x) * RL_TW;
        int tide = tid;                             // opaque per tile: keeps the epilogue address math out of the registers
        asm volatile("" : "+v"(tide));              // that live across the K loops
        const int lane_e = tide & 63, wave_e = tide >> 6, q_e = lane_e >> 4, lp_e = rv_pix16(lane_e & 15);

        // ---------------- phase 1: acc1 = conv1(x) on the halo region ----------------------------------------------
        f32x4 acc1[MT][RL_T1W];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < RL_T1W; ++t) acc1[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rl_kloop<MT, RL_T1W>(acc1, wl1, tab1 + q, xt, pb1, p.S, lane);
        if (iter == p.probe_iter) RL_STAMP(3);
        // residual x of this lane's phase-2 outputs: the x tile is about to be overwritten by t
        f16x4 xres[MT][RL_T2W];
#pragma unroll
        for (int t = 0; t < RL_T2W; ++t) {
            const int ti = wave_e * RL_T2W + t;
            const unsigned char* xr_ = xt + (size_t)(((ti >> 1) + 2) * RL_XW + (ti & 1) * 16 + lp_e + 2) * psb;
#pragma unroll
            for (int m = 0; m < MT; ++m) xres[m][t] = *reinterpret_cast<const f16x4*>(xr_ + min(m * 16 + q_e * 4, p.c - 4) * 2);
        }
        __syncthreads();                            // A: every wave is done reading the x tile
        // t = act(acc1 + b1), zero outside the frame, over the x tile.  (This epilogue costs as many cycles as the K loop
        // before it -- 16 waves share four vector ALUs -- so it is written for instruction count: the ReLU blocks, 128 of the
        // 156 fused launches of a RefVSR_small frame, skip the slope multiply, and the fp16 pairs are converted with two
        // packed conversions -- through a 4-vector + mask union hipcc emits two scalar conversions and a permute.)
        auto epi1 = [&](auto relu_c) {
            constexpr bool RELU = decltype(relu_c)::value;
#pragma unroll
            for (int t = 0; t < RL_T1W; ++t) {
                const int tile1 = wave_e * RL_T1W + t;
                const int pix = tile1 * 16 + lp_e;
                if (tile1 >= RL_T1 || pix >= RL_NI) continue;
                const int r = (int)(((float)pix + 0.5f) * (1.0f / (float)RL_IW));
                const int iy = ty0 - 1 + r, ix = tx0 - 1 + (pix - r * RL_IW);
                const unsigned keep = (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) ? 0xffffffffu : 0u;
                unsigned char* dst = xt + (size_t)pix * psb;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int co0 = m * 16 + q_e * 4;
                    if (co0 >= p.c) continue;
                    const float4 bv = *reinterpret_cast<const float4*>(bl + m * 16 + q_e * 4);
                    float y0 = acc1[m][t][0] + bv.x, y1 = acc1[m][t][1] + bv.y, y2 = acc1[m][t][2] + bv.z, y3 = acc1[m][t][3] + bv.w;
                    if constexpr (RELU) {
                        y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); y2 = fmaxf(y2, 0.f); y3 = fmaxf(y3, 0.f);
                    } else {                                                                   // leaky ReLU, 0 < slope <= 1
                        y0 = fmaxf(y0, y0 * p.act_slope); y1 = fmaxf(y1, y1 * p.act_slope);
                        y2 = fmaxf(y2, y2 * p.act_slope); y3 = fmaxf(y3, y3 * p.act_slope);
                    }
                    union { f16x2 h; unsigned u; } a, b;
                    a.h = (f16x2){(f16)y0, (f16)y1};
                    b.h = (f16x2){(f16)y2, (f16)y3};
                    *reinterpret_cast<uint2*>(dst + co0 * 2) = make_uint2(a.u & keep, b.u & keep);
                }
            }
        };
        if (p.act_slope == 0.f) epi1(std::true_type{}); else epi1(std::false_type{});
        if (has_next) x_fetch(tl + k_step);         // in flight during conv2
        if (iter == p.probe_iter) RL_STAMP(4);
        __syncthreads();                            // B: t complete
        if (iter == p.probe_iter) RL_STAMP(5);

        // ---------------- phase 2: out = post(x + conv2(t) + b2) ----------------------------------------------------
        f32x4 acc2[MT][RL_T2W];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < RL_T2W; ++t) acc2[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rl_kloop<MT, RL_T2W>(acc2, wl2, tab2 + q, xt, pb2, p.S, lane);
        if (iter == p.probe_iter) RL_STAMP(6);
        if (has_next) {
            __syncthreads();                        // C: every wave is done reading t
            x_park();
        }
#pragma unroll
        for (int t = 0; t < RL_T2W; ++t) {
            const int ti = wave_e * RL_T2W + t;
            const int oy = ty0 + (ti >> 1), ox = tx0 + (ti & 1) * 16 + lp_e;
            if (oy >= p.h || ox >= p.w) continue;
            f16* dst = p.out + ((size_t)oy * p.w + ox) * p.c;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = m * 16 + q_e * 4;
                if (co0 >= p.c) continue;
                const float4 bv = *reinterpret_cast<const float4*>(bl + 32 + m * 16 + q_e * 4);
                const f16x4 xv4 = xres[m][t];
                float y[4] = {acc2[m][t][0] + bv.x + (float)xv4[0], acc2[m][t][1] + bv.y + (float)xv4[1],
                              acc2[m][t][2] + bv.z + (float)xv4[2], acc2[m][t][3] + bv.w + (float)xv4[3]};
                if (p.post_slope != 1.0f) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = fmaxf(y[i], y[i] * p.post_slope);
                }
                f16x4 o = {(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
                *reinterpret_cast<f16x4*>(dst + co0) = o;
            }
        }
        if (iter == p.probe_iter) RL_STAMP(7);
        if (has_next) __syncthreads();              // D: next x tile visible
    }
    if (p.probe && tid == 0) {
        p.probe[blockIdx.x * 12 + 9] = __builtin_amdgcn_s_memrealtime();
        p.probe[blockIdx.x * 12 + 10] = __builtin_amdgcn_s_memtime();
    }
#undef RL_STAMP
}

extern unsigned long long* g_rb_probe;             // runtime.hip: refvsr_set_probe
extern int g_rb_probe_iter;

static size_t rl_lds_bytes(int c) {
    const int ncg = c / 8, ps = ncg | 1;
    const int S = rv_ksteps(3, ncg);
    const int MT = (c + 15) / 16;
    return (size_t)((S * 4 * 2 * 4 + 256 + 15) / 16 * 16) + 2 * (size_t)S * MT * 2 * 1024 + (size_t)RL_XH * RL_XW * ps * 16;
}

extern "C" int refvsr_resblock_lean_fits(int c) {
    if (c <= 0 || c % 8 != 0) return 0;
    if ((c + 15) / 16 > 2) return 0;
    if (RL_XW * (c / 8) > 256) return 0;
    return rl_lds_bytes(c) <= 160 * 1024 ? 1 : 0;
}

static int g_lean_waves = 8;               // A/B knob (refvsr_set_resblock_waves): 4 or 8 waves per workgroup
extern "C" int refvsr_set_resblock_waves(int waves) {
    if (waves != 4 && waves != 8) return 1;
    g_lean_waves = waves;
    return 0;
}

template <int MT, bool WIDE, int NWV>
static int launch_lean(ResLeanArgs& a, size_t lds, hipStream_t st) {
    // per device: the dynamic-LDS attribute and the occupancy (a process may drive several GPUs)
    static bool attr_done[RV_MAX_DEVICES] = {};
    static int occ_dev[RV_MAX_DEVICES] = {};
    static size_t occ_lds[RV_MAX_DEVICES] = {};
    const int dev = rv_device();
    if (!attr_done[dev]) {
        RV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_lean_kernel<MT, WIDE, NWV>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[dev] = true;
    }
    if (occ_dev[dev] == 0 || occ_lds[dev] != lds) {
        int occ = 0;
        RV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, resblock_lean_kernel<MT, WIDE, NWV>, NWV * 64, lds));
        occ_dev[dev] = occ < 1 ? 1 : occ;
        occ_lds[dev] = lds;
    }
    int cap = (rv_stream_cus(st) * occ_dev[dev]) & ~7;
    if (cap < 8) cap = 8;
    const int gx = a.n_tiles < cap ? a.n_tiles : cap;
    a.grid = gx;
    hipLaunchKernelGGL((resblock_lean_kernel<MT, WIDE, NWV>), dim3(gx), dim3(NWV * 64), lds, st, a);
    RV_LAUNCH_CHECK();
    return 0;
}

extern "C" int refvsr_resblock_lean(const void* src, int c, int h, int w, const void* w1, const float* b1,
                                    const void* w2, const float* b2, int ksteps, float act_slope, float post_slope,
                                    void* out, void* stream) {
    RV_CHECK(src && out && w1 && w2 && b1 && b2 && h > 0 && w > 0, "resblock_lean: bad args");
    RV_CHECK(src != out, "resblock_lean: in-place operation is not supported (neighbouring tiles read the input halo)");
    RV_CHECK(refvsr_resblock_lean_fits(c), "resblock_lean: channel count %d not supported by the fused kernel", c);
    RV_CHECK(act_slope >= 0.f && act_slope <= 1.f && post_slope >= 0.f && post_slope <= 1.f,
             "resblock_lean: activation slopes must lie in [0, 1]");
    RV_CHECK(refvsr_init() == 0, "init failed");
    ResLeanArgs a;
    memset(&a, 0, sizeof(a));
    a.src = (const f16*)src; a.out = (f16*)out;
    a.c = c; a.ncg = c / 8; a.ps = a.ncg | 1; a.h = h; a.w = w;
    a.G = 9 * a.ncg; a.S = rv_ksteps(3, a.ncg);
    RV_CHECK(a.S == ksteps, "resblock_lean: ksteps mismatch (%d vs %d)", ksteps, a.S);
    a.inv_ncg = 1.0f / (float)a.ncg;
    a.w1 = (const uint4*)w1; a.b1 = b1; a.w2 = (const uint4*)w2; a.b2 = b2;
    a.act_slope = act_slope; a.post_slope = post_slope;
    a.probe = g_rb_probe; a.probe_iter = g_rb_probe_iter;
    const int MT = (c + 15) / 16;
    a.tab_bytes = (a.S * 4 * 2 * 4 + 256 + 15) / 16 * 16;         // two K-slot tables + 2 x 32 bias floats
    a.w_bytes = a.S * MT * 2 * 1024;
    a.tiles_x = rv_cdiv(w, RL_TW);
    a.n_tiles = a.tiles_x * rv_cdiv(h, RL_TH);
    const size_t lds = rl_lds_bytes(c);
    hipStream_t st = (hipStream_t)stream;
    const bool wide = RL_XW * a.ncg > 128;
#define RL_CASE(M, WD, NW_)                                                          \
    if (MT == M && wide == WD && g_lean_waves == NW_) return launch_lean<M, WD, NW_>(a, lds, st);
    RL_CASE(1, false, 8) RL_CASE(2, false, 8) RL_CASE(1, true, 8) RL_CASE(2, true, 8)
    RL_CASE(1, false, 4) RL_CASE(2, false, 4) RL_CASE(1, true, 4) RL_CASE(2, true, 4)
#undef RL_CASE
    refvsr_set_error("resblock_lean: no kernel for MT=%d wide=%d waves=%d", MT, (int)wide, g_lean_waves);
    return 1;
}

// A run of n fused blocks x <- post(x + conv2(act(conv1 x))) behind ONE call: n launches of the kernel above on the caller's
// stream, intermediate maps ping-ponging between two scratch buffers (the blocks cannot run in place: neighbouring tiles read
// the input halo).  Same results as n calls of refvsr_resblock_lean; what it saves is the host: one FFI crossing and two
// allocations instead of n of each (156 of the ~330 launches of a RefVSR_small frame are fused blocks).
extern "C" int refvsr_resblock_chain(const void* src, int c, int h, int w, int n, const void* const* w1,
                                     const float* const* b1, const void* const* w2, const float* const* b2, int ksteps,
                                     float act_slope, float post_slope, void* scratch0, void* scratch1, void* out,
                                     void* stream) {
    RV_CHECK(n >= 1 && w1 && b1 && w2 && b2 && out, "resblock_chain: bad args");
    RV_CHECK(n == 1 || scratch0, "resblock_chain: n >= 2 needs scratch0");
    RV_CHECK(n <= 2 || scratch1, "resblock_chain: n >= 3 needs scratch1");
    RV_CHECK(src != out && scratch0 != out && scratch1 != out && (n < 2 || scratch0 != src) && (n < 3 || scratch1 != src) &&
             (n < 3 || scratch0 != scratch1), "resblock_chain: buffers must be distinct");
    const void* cur = src;
    for (int i = 0; i < n; ++i) {
        void* dst = (i == n - 1) ? out : ((i & 1) ? scratch1 : scratch0);
        if (refvsr_resblock_lean(cur, c, h, w, w1[i], b1[i], w2[i], b2[i], ksteps, act_slope, post_slope, dst, stream)) return 1;
        cur = dst;
    }
    return 0;
}

