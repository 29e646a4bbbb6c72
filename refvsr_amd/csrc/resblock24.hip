// Fused residual block for the C = 24 maps of RefVSR_small:  out = x + conv2( act( conv1(x) ) ),  3x3, 24 -> 24,
// with EVERYTHING a compile-time constant.  Replaces the per-block body of ResidualBlockNoBN
// (mmedit/models/common/sr_backbone_utils.py:42-97: 96 launches per frame in backward/forward_resblocks,
// models/archs/RefVSR.py:327-360) and of ResBlock (models/archs/RefVSR_/common.py:25-39: 60 launches per frame in the
// ResLists of AA_AF_conf_prop / compute_up).
//
// Why a second kernel next to resblock_lean.hip (which stays as the runtime-generic kernel for C = 8 / 16 / 32): the SQ
// counters of the round-2 kernel (profiles/r02_pmc_sq_match_top2.txt) show it bound by INSTRUCTION ISSUE, not by the matrix
// pipe or by memory -- per wave and tile 609 vector-ALU + 166 LDS + 140 MFMA instructions, 26 % of the wave cycles issuing
// x 4 waves per SIMD = a saturated issue port, matrix pipe 19-36 % busy.  Most of those instructions are index arithmetic
// that exists only because channel count, K order and tile geometry were runtime values.  Here:
//
//   * K order chosen so that a B-fragment address is  (per-lane base) + (immediate):  the K-blocks of one 3x3 window row are
//     nine consecutive 16-byte slots u = 3*tx + cg of the x tile (pixel stride = 3 slots, no padding slot), K-step
//     s = 2*ty + a (a = 0|1) takes u = 4a + {0,2,1,3}[q] (q = lane >> 4: the two K-blocks a ds_read_b128 lane group mixes
//     have slot offsets of equal parity => bank-conflict free, common.h), K-step 6 takes u = 8 of rows ty = q (q = 3: a zero
//     block).  No K-slot table, no per-step address arithmetic: a K-step is 3 + T ds_read_b128 and 3T MFMAs, nothing else;
//   * hi + lo weights in THREE 16-row fragments instead of four: rows [hi 0-15], [lo 0-15], [hi 16-23 | lo 16-23] -- the
//     third accumulator holds the hi sums of channels 16-23 in lanes 0-31 and their lo sums in lanes 32-63, folded with one
//     v_permlane32_swap + add per register after the K loop (25 % fewer MFMAs and weight-fragment reads, 42 KB of weights);
//   * bias (and, for conv2, bias + residual) are the accumulators' initial values; ReLU runs on the packed fp16 pair;
//   * the x tile (12 x 36 pixels x 48 bytes) is a contiguous image of 12 row segments of the HWC map: three 16-byte loads
//     per thread at offsets computed once per kernel, zero masking only in tiles that touch the frame border (uniform branch);
//   * weights + biases of a block are ONE 43 264-byte blob that goes global -> LDS with global_load_lds_dwordx4.
//
// Layout of the blob (host: refvsr_amd/packing.py:pack_resblock24):  [conv1: 7 K-steps x 3 fragments x 1 KiB][conv2: same]
// [b1: 32 floats, 24..31 = 0][b2: 32 floats].  Fragment f of K-step s, lane l = (q, r): 8 halfs = K-block (s, q) of row r.
//
//   stage   x tile                                                            global -> registers -> LDS
//   phase 1 acc1 = b1 + conv1(x) on the 10 x 34 halo region (22 sixteen-pixel groups over the waves), residual x values
//           -> registers;  barrier;  t = act(acc1) (0 outside the frame) -> LDS over the x tile, shifted by (1, 1);  barrier
//   phase 2 out = (b2 + x) + conv2(t) on the 8 x 32 tile -> global (8-byte HWC channel vectors)
#include <type_traits>

#include "common.h"

#ifndef REFVSR_RB24_STORE_DEFAULT
#define REFVSR_RB24_STORE_DEFAULT 1
#endif
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int RB_TW = 32;                                 // tile width; the tile height TH is a template parameter (8 | 16)
constexpr int RB_XW = RB_TW + 4;                          // x tile: (TH + 4) x 36 pixels
constexpr int RB_IW = RB_TW + 2;                          // intermediate: (TH + 2) x 34
constexpr int RB_PXB = 48;                                // bytes per pixel (24 halfs)
constexpr int RB_ROWB = RB_XW * RB_PXB;                   // 1728 bytes per tile row
constexpr int RB_RCH = RB_ROWB / 16;                      // 108 chunks per tile row
constexpr int rb_xbytes(int th) { return (th + 4) * RB_ROWB; }         // 20736 (TH = 8) | 34560 (TH = 16)
constexpr int RB_S = 7, RB_NF = 3;
constexpr int RB_WB = RB_S * RB_NF * 1024;                // 21504 bytes of fragments per conv
constexpr int RB_BIAS = 2 * RB_WB;                        // 43008
constexpr int RB_BLOB = RB_BIAS + 256;                    // 43264
constexpr int RB_XT = RB_BLOB;                            // LDS offset of the x tile
constexpr int rb_lds(int th) { return RB_XT + rb_xbytes(th); }         // 64000 -> two workgroups per CU | 77824 -> one
static_assert(RB_BLOB == REFVSR_RESBLOCK24_BLOB_BYTES, "blob size is part of the C-ABI");
}  // namespace

struct RB24Args {
    const unsigned char* src; unsigned char* out; const unsigned char* blob;
    int h, w, tiles_x, n_tiles, grid;
    float act_slope;
    unsigned long long* probe;           // PROBE kernels: per-workgroup s_memtime stamps (refvsr_set_probe), 12 per workgroup
    int probe_iter;                      // which tile iteration of the workgroup is stamped
    // HEAD kernels (refvsr_conv_hr_last): `out` is planar fp32 [3][h][w]; base_lr = the LR centre frame, planar fp32 [3][bh][bw]
    const float* base_lr; int bh, bw; float base_step;
    int out_fmt;                         // HEAD kernels: REFVSR_RESULT_* of `out` (planar [3][h][w])
    // Multi-map launches (refvsr_resblock24_chain_batch): batch > 1 maps of one geometry share the launch and the weight fill; the
    // flat tile index t = b * tpm + (tile of map b), map b reads bsrc[b] and writes bout[b].  batch <= 1: src / out above.
    int batch, tpm;
    const unsigned char* bsrc[REFVSR_MAX_MAPS]; unsigned char* bout[REFVSR_MAX_MAPS];
};

// lane l: a[l] + a[l ^ 32]   (v_mov, v_permlane32_swap, v_add per register).  The two results are taken out of the builtin's
// vector as plain unsigned values first: __builtin_bit_cast on `pr[1]` directly made hipcc (ROCm 7.2) read element 0 twice.
__device__ __forceinline__ float rb_fold1(const float a) {
    const unsigned u = __float_as_uint(a);
    const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned x0 = pr[0], x1 = pr[1];
    return __uint_as_float(x0) + __uint_as_float(x1);
}
__device__ __forceinline__ f32x4 rb_fold_halves(const f32x4 a) {
    f32x4 r;
    r[0] = rb_fold1(a[0]); r[1] = rb_fold1(a[1]); r[2] = rb_fold1(a[2]); r[3] = rb_fold1(a[3]);
    return r;
}

// K loop of one conv: T pixel groups of this wave, fragments at smem + wofs, B windows at smem + pb[t] (+ immediates).
// Software pipelined over two fragment sets (the reads of step s+1 are issued above the MFMAs of step s).
template <int T, int TA>
__device__ __forceinline__ void rb_kloop(f32x4 (&acc0)[TA], f32x4 (&acc1)[TA], const unsigned char* smem, const int wofs,
                                         const int la, const int (&pb)[TA], const int delta6) {
    static_assert(T <= TA, "group count");
    uint4 a[2][RB_NF], b[2][T];
    auto load = [&](auto sc, uint4 (&af)[RB_NF], uint4 (&bf)[T]) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int f = 0; f < RB_NF; ++f) af[f] = *reinterpret_cast<const uint4*>(smem + wofs + (s * RB_NF + f) * 1024 + la);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if constexpr (s < 6) bf[t] = *reinterpret_cast<const uint4*>(smem + pb[t] + (s >> 1) * RB_ROWB + (s & 1) * 64);
            else bf[t] = *reinterpret_cast<const uint4*>(smem + pb[t] + delta6);
        }
    };
    auto mfma = [&](const uint4 (&af)[RB_NF], const uint4 (&bf)[T]) {
        const f16x8 a_hi = *reinterpret_cast<const f16x8*>(&af[0]);
        const f16x8 a_lo = *reinterpret_cast<const f16x8*>(&af[1]);
        const f16x8 a_mx = *reinterpret_cast<const f16x8*>(&af[2]);
#pragma unroll
        for (int t = 0; t < T; ++t) acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, *reinterpret_cast<const f16x8*>(&bf[t]), acc0[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_mx, *reinterpret_cast<const f16x8*>(&bf[t]), acc1[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, *reinterpret_cast<const f16x8*>(&bf[t]), acc0[t], 0, 0, 0);
    };
    load(std::integral_constant<int, 0>{}, a[0], b[0]);
    load(std::integral_constant<int, 1>{}, a[1], b[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[0], b[0]);
    load(std::integral_constant<int, 2>{}, a[0], b[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[1], b[1]);
    load(std::integral_constant<int, 3>{}, a[1], b[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[0], b[0]);
    load(std::integral_constant<int, 4>{}, a[0], b[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[1], b[1]);
    load(std::integral_constant<int, 5>{}, a[1], b[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[0], b[0]);
    load(std::integral_constant<int, 6>{}, a[0], b[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[1], b[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[0], b[0]);
}

// K loop of the output head's conv (HEAD kernels): ONE fragment per K-step -- rows 0-2 = hi, rows 8-10 = lo of the three output
// channels -- at the blob's fragment slot (s, 0); the other two slots of the 3-fragment layout are not read.
template <int T>
__device__ __forceinline__ void rb_kloop_head(f32x4 (&acc0)[T], const unsigned char* smem, const int wofs, const int la,
                                              const int (&pb)[T], const int delta6) {
    uint4 a[2], b[2][T];
    auto load = [&](auto sc, uint4& af, uint4 (&bf)[T]) {
        constexpr int s = decltype(sc)::value;
        af = *reinterpret_cast<const uint4*>(smem + wofs + s * RB_NF * 1024 + la);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if constexpr (s < 6) bf[t] = *reinterpret_cast<const uint4*>(smem + pb[t] + (s >> 1) * RB_ROWB + (s & 1) * 64);
            else bf[t] = *reinterpret_cast<const uint4*>(smem + pb[t] + delta6);
        }
    };
    auto mfma = [&](const uint4& af, const uint4 (&bf)[T]) {
        const f16x8 a_w = *reinterpret_cast<const f16x8*>(&af);
#pragma unroll
        for (int t = 0; t < T; ++t) acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_w, *reinterpret_cast<const f16x8*>(&bf[t]), acc0[t], 0, 0, 0);
    };
    load(std::integral_constant<int, 0>{}, a[0], b[0]);
    load(std::integral_constant<int, 1>{}, a[1], b[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[0], b[0]);
    load(std::integral_constant<int, 2>{}, a[0], b[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[1], b[1]);
    load(std::integral_constant<int, 3>{}, a[1], b[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[0], b[0]);
    load(std::integral_constant<int, 4>{}, a[0], b[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[1], b[1]);
    load(std::integral_constant<int, 5>{}, a[1], b[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[0], b[0]);
    load(std::integral_constant<int, 6>{}, a[0], b[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[1], b[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma(a[0], b[0]);
}

// act(y) of four fp32 values -> two packed fp16 pairs.  RELU: conversion first, v_pk_max_f16 on the pairs (ReLU commutes with
// the rounding); leaky ReLU (0 < slope <= 1): max(y, slope * y) in fp32.
template <bool RELU>
__device__ __forceinline__ uint2 rb_act_pack(const f32x4 y, const float slope) {
    union { f16x2 h; unsigned u; } a, b;
    if constexpr (RELU) {
        const f16x2 z = {(f16)0.f, (f16)0.f};
        a.h = __builtin_elementwise_max((f16x2){(f16)y[0], (f16)y[1]}, z);
        b.h = __builtin_elementwise_max((f16x2){(f16)y[2], (f16)y[3]}, z);
    } else {
        a.h = (f16x2){(f16)fmaxf(y[0], y[0] * slope), (f16)fmaxf(y[1], y[1] * slope)};
        b.h = (f16x2){(f16)fmaxf(y[2], y[2] * slope), (f16)fmaxf(y[3], y[3] * slope)};
    }
    return make_uint2(a.u, b.u);
}

__device__ __forceinline__ uint2 rb_pack(const f32x4 y) {
    union { f16x2 h; unsigned u; } a, b;
    a.h = (f16x2){(f16)y[0], (f16)y[1]};
    b.h = (f16x2){(f16)y[2], (f16)y[3]};
    return make_uint2(a.u, b.u);
}

// NWV = waves per workgroup (8: three + two pixel groups per wave, <= 128 VGPRs, four waves per SIMD with the two workgroups
// of a CU; 4: six + four groups per wave, twice the weight-fragment reuse, two waves per SIMD).
// STORE: how the output tile leaves the workgroup.  0 (round 3): 8-byte stores, lane (q, pixel) writes its own four channels.
// 1: 16-byte stores -- the lanes q and q ^ 1 exchange halves with v_permlane16_swap so that a lane holds EIGHT consecutive
// channels of one pixel (of the wave's left 16-pixel group for even q, of the right one for odd q): half the store instructions
// (the store phase is issue bound: 8 waves x 6 stores of 8 bytes per tile).  Measured (profiles/r04_resblock_microbench.txt, us per
// block 0 / 1): LR 9.17 / 9.19, 2x 28.73 / 27.80, HR 103.3 / 102.2 -> 1 is the default.  (Round 4 also carried the same stores as
// write-through `sc1` stores: slower everywhere -- 9.47 / 29.68 / 107.6 us, 197.1 vs 199.5 frames/s -- removed in round 5.)
// HEAD = 1 (refvsr_conv_hr_last, round 4): not a residual block but the last two convs of the upsampler, conv_hr (24 -> 24,
// LeakyReLU 0.1) and conv_last (24 -> 3) + the bicubic base + the clamps (RefVSR.py:91-92,116-118,288,297), on this kernel's
// two-conv skeleton: conv1 = conv_hr, the intermediate tile stays in LDS (the 100 MB HR map between the two convs never exists),
// conv2 = ONE fragment per K-step (rows 0-2 hi, rows 8-10 lo), no residual, the epilogue of refvsr_conv_last (fold, one lane per
// channel, rv_bicubic_at, planar fp32 stores).  The blob keeps the block layout (conv2's fragment slots 1 and 2 unused).
template <bool RELU, int NWV, bool PROBE = false, int TH = 8, int STORE = 0, int HEAD = 0>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(NWV == 4 ? 2 : 4, NWV == 4 ? 2 : 4))) void resblock24_kernel(RB24Args p) {
    static_assert((TH == 8 && (NWV == 4 || NWV == 8)) || (TH == 16 && NWV == 16), "tile height / waves");
    static_assert(HEAD == 0 || (!RELU && !PROBE && NWV != 4), "output-head variant");
    constexpr int RB_TH = TH, RB_XH = TH + 4, RB_IH = TH + 2;    // TH = 8: x tile 12 x 36, intermediate 10 x 34 (22 sixteen-pixel groups)
    constexpr int RB_NI = RB_IH * RB_IW;                         // TH = 16: 20 x 36, 18 x 34 (39 groups), 16 waves, ONE workgroup per CU
    constexpr int RB_G1 = (RB_NI + 15) / 16;                     //   (the large maps, see refvsr_resblock24_chain)
    constexpr int RB_XBYTES = rb_xbytes(TH), RB_XCH = RB_XBYTES / 16;
    // PROBE: stamps 0 entry, 1 prologue loads issued, 2 first barrier passed; of tile `probe_iter`: 3 conv1 K loop done, 4 fold +
    // barrier A, 5 t written (+ next tile's loads issued), 6 barrier B, 7 conv2 K loop done, 8 barrier C + next tile parked,
    // 9 stores issued, 10 barrier D; 11 exit ([8]/[10] = previous stamp when there is no next tile)
#define RB_STAMP(i) do { if constexpr (PROBE) { if (p.probe && threadIdx.x == 0) p.probe[blockIdx.x * 12 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
    RB_STAMP(0);
    constexpr int NT = NWV * 64;
    constexpr int T1 = (RB_G1 + NWV - 1) / NWV;                  // phase-1 groups of a "full" wave (3 | 6)
    constexpr int T1REM = RB_G1 % NWV;                           // waves below this index are full, the others have T1 - 1
    constexpr int T2 = 2 * TH / NWV;                             // phase-2 groups per wave (2 | 4)
    constexpr int KCH = (RB_XCH + NT - 1) / NT;                  // x-tile chunks per thread (3 | 6)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    asm volatile("" :: "s"(p.src), "s"(p.out), "s"(p.blob), "s"(p.h), "s"(p.w), "s"(p.tiles_x), "s"(p.n_tiles), "s"(p.grid),
                 "s"(p.act_slope));                             // one batch of scalar argument loads
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weights + biases: global -> LDS, 1 KiB per wave instruction, issued before anything else --------------------------
    {
        constexpr int NCH = RB_BIAS / 1024;                      // 42 full chunks + the 256-byte bias tail
        const unsigned char* g = p.blob + lane * 16;
#pragma unroll
        for (int j = 0; j < (NCH + NWV - 1) / NWV; ++j) {
            const int c = wave + j * NWV;
            if (c < NCH)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + c * 1024),
                                                 (__attribute__((address_space(3))) void*)(smem + c * 1024), 16, 0, 0);
        }
        if (wave == NCH % NWV && lane < 16)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + NCH * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + NCH * 1024), 16, 0, 0);
    }

    const int rowb_g = p.w * RB_PXB;                             // bytes per row of the HWC maps
    // ---- first: the loads.  x-tile chunk k of this thread = 16 bytes at tile row r, row chunk cr (i = tid + k NT = r 108 + cr):
    // global offset relative to the tile origin, computed once (one division, then i += NT = A 108 + B steps)
    unsigned xg[KCH];
    {
        constexpr int A = NT / RB_RCH, B = NT % RB_RCH;
        int r = tid / RB_RCH, cr = tid - (tid / RB_RCH) * RB_RCH;
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            xg[k] = (unsigned)(min(r, RB_XH - 1) * rowb_g + cr * 16);     // (threads past the tile: a valid address, value dropped)
            const bool wrap = cr + B >= RB_RCH;
            r += A + (wrap ? 1 : 0);
            cr += B - (wrap ? RB_RCH : 0);
        }
    }
    uint4 xv[KCH];
    // tile origin (top-left pixel of the x tile) may lie outside the frame: only in-frame chunks are dereferenced
    auto x_fetch = [&](const int tf) {
        int t = tf;
        const unsigned char* srcp = p.src;
        if (p.batch > 1) {                                       // flat tile index -> (map, tile of the map); uniform
            const int bm = (int)((unsigned)tf / (unsigned)p.tpm);
            t = tf - bm * p.tpm;
            srcp = p.bsrc[bm];
        }
        const int tyi = t / p.tiles_x;
        const int ty0 = tyi * RB_TH, tx0 = (t - tyi * p.tiles_x) * RB_TW;
        const bool interior = ty0 >= 2 && ty0 + RB_TH + 2 <= p.h && tx0 >= 2 && tx0 + RB_TW + 2 <= p.w;
        const long long org = ((long long)(ty0 - 2) * p.w + (tx0 - 2)) * RB_PXB;
        if (interior) {
            const unsigned char* b = srcp + org;
#pragma unroll
            for (int k = 0; k < KCH; ++k) xv[k] = *reinterpret_cast<const uint4*>(b + xg[k]);
        } else {
            // (thread id made opaque: hipcc otherwise hoists this path's index arithmetic out of the tile loop and keeps it
            //  in registers that live across the K loops)
            int tide = tid;
            asm volatile("" : "+v"(tide));
#pragma unroll
            for (int k = 0; k < KCH; ++k) {
                const int i = min(tide + k * NT, RB_XCH - 1);
                const int r = i / RB_RCH;
                const int px = (i - r * RB_RCH) / 3;
                const int iy = ty0 - 2 + r, ix = tx0 - 2 + px;
                const bool ok = (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w;
                const unsigned off = (unsigned)((min(max(iy, 0), p.h - 1) * p.w + min(max(ix, 0), p.w - 1)) * RB_PXB + ((i - r * RB_RCH) - px * 3) * 16);
                uint4 v = *reinterpret_cast<const uint4*>(srcp + off);           // clamped address, masked value (32-bit offsets: host check)
                const unsigned keep = ok ? 0xffffffffu : 0u;
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                xv[k] = v;
            }
        }
    };
    auto x_park = [&]() {
#pragma unroll
        for (int k = 0; k < KCH; ++k)
            if (k * NT + NT <= RB_XCH || tid + k * NT < RB_XCH)
                *reinterpret_cast<uint4*>(smem + RB_XT + tid * 16 + k * NT * 16) = xv[k];
    };

    int tl, k_hi;
    rv_tile_range(p.n_tiles, p.grid, tl, k_hi);
    if (tl < k_hi) x_fetch(tl);
    __builtin_amdgcn_sched_barrier(0);                           // weights and first tile in flight before the rest of the set-up
    RB_STAMP(1);

    // ---- per-lane constants ---------------------------------------------------------------------------------------------
    const int q = lane >> 4;
    const int lp = rv_pix16(lane & 15);                          // pixel of a 16-pixel group held by this lane's MFMA column
    const int permq = ((q & 1) << 1) | (q >> 1);                 // {0, 2, 1, 3}
    const int la = lane * 16;
    const int delta6 = min(q, 2) * RB_ROWB + 128 - permq * 16;   // K-step 6 (u = 8 of window row q) relative to a window base
    const int dq = RB_ROWB + RB_PXB + q * 8 - permq * 16;        // channels 4q.. of the window's centre pixel, same base
    const bool full1 = T1REM == 0 || wave < T1REM;
    const int g1 = full1 ? wave * T1 : T1REM * T1 + (wave - T1REM) * (T1 - 1);
    int pb1[T1];                                                 // phase 1: LDS byte offset of each group's window origin (+ permq slot)
#pragma unroll
    for (int t = 0; t < T1; ++t) {
        const int pix = min((g1 + t) * 16 + lp, RB_NI - 1);      // lanes past the region repeat its last pixel (same value, same address)
        const int r = pix / RB_IW;
        pb1[t] = RB_XT + r * RB_ROWB + (pix - r * RB_IW) * RB_PXB + permq * 16;
    }
    const int oy0 = (wave * T2) >> 1;                            // first output row of this wave
    int pb2[T2];                                                 // phase 2 windows: one VGPR + immediates
#pragma unroll
    for (int t = 0; t < T2; ++t)
        pb2[t] = RB_XT + (oy0 + (t >> 1) + 1) * RB_ROWB + ((t & 1) * 16 + lp + 1) * RB_PXB + permq * 16;
    const unsigned oo = (unsigned)(oy0 * rowb_g + lp * RB_PXB + q * 8);   // output offset of group 0 relative to the tile origin
    if (tl < k_hi) x_park();
    __syncthreads();                                             // weights, biases, first tile
    RB_STAMP(2);

    for (int iter = 0; tl < k_hi; ++tl, ++iter) {
        const bool stamp = PROBE && iter == p.probe_iter;
        const bool has_next = tl + 1 < k_hi;
        int tm = tl;
        unsigned char* outp = p.out;
        if (p.batch > 1) {
            const int bm = (int)((unsigned)tl / (unsigned)p.tpm);
            tm = tl - bm * p.tpm;
            outp = p.bout[bm];
        }
        const int tyi = tm / p.tiles_x;
        const int ty0 = tyi * RB_TH, tx0 = (tm - tyi * p.tiles_x) * RB_TW;
        const bool interior = ty0 >= 2 && ty0 + RB_TH + 2 <= p.h && tx0 >= 2 && tx0 + RB_TW + 2 <= p.w;

        // ---------------- phase 1: acc = b1 + conv1(x) on the halo region -------------------------------------------------
        f32x4 a0[T1], a1[T1];
        {
            const f32x4 bv0 = *reinterpret_cast<const f32x4*>(smem + RB_BIAS + q * 16);          // channels 4q ..
            const f32x4 bv1 = *reinterpret_cast<const f32x4*>(smem + RB_BIAS + 64 + q * 16);     // channels 16 + 4q .. (24..31: zeros)
#pragma unroll
            for (int t = 0; t < T1; ++t) { a0[t] = bv0; a1[t] = bv1; }
        }
        if (full1) rb_kloop<T1, T1>(a0, a1, smem, 0, la, pb1, delta6);
        else rb_kloop<T1 - 1, T1>(a0, a1, smem, 0, la, pb1, delta6);
        if (stamp) RB_STAMP(3);
        if (has_next) x_fetch(tl + 1);                           // next tile: in flight from here to the end of conv2
        // residual x values of this lane's outputs: the x tile is about to be overwritten by t
        f16x4 xr0[T2], xr1[T2];
        if constexpr (HEAD == 0) {
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                xr0[t] = *reinterpret_cast<const f16x4*>(smem + pb2[t] + dq);
                // lanes 32-63 of the third accumulator collect lo sums only: they start from zeros (the pad of b1)
                xr1[t] = *reinterpret_cast<const f16x4*>(smem + (q < 2 ? pb2[t] + dq + 32 : RB_BIAS + 96));
            }
        }
#pragma unroll
        for (int t = 0; t < T1; ++t) a1[t] = rb_fold_halves(a1[t]);
        __syncthreads();                                         // A: every wave is done reading the x tile
        if (stamp) RB_STAMP(4);
        // t = act(acc), zero outside the frame (conv2's zero padding), written over the x tile at (+1, +1)
        auto epi1 = [&](auto tc) {
            constexpr int T = decltype(tc)::value;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                uint2 v0 = rb_act_pack<RELU>(a0[t], p.act_slope);
                uint2 v1 = rb_act_pack<RELU>(a1[t], p.act_slope);
                if (!interior) {
                    int lpe = lp;
                    asm volatile("" : "+v"(lpe));                 // see x_fetch
                    const int pix = min((g1 + t) * 16 + lpe, RB_NI - 1);
                    const int r = pix / RB_IW;
                    const int iy = ty0 - 1 + r, ix = tx0 - 1 + (pix - r * RB_IW);
                    const unsigned keep = ((unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w) ? 0xffffffffu : 0u;
                    v0.x &= keep; v0.y &= keep; v1.x &= keep; v1.y &= keep;
                }
                unsigned char* d = smem + pb1[t] + dq;
                *reinterpret_cast<uint2*>(d) = v0;
                if (q < 2) *reinterpret_cast<uint2*>(d + 32) = v1;
            }
        };
        if (full1) epi1(std::integral_constant<int, T1>{}); else epi1(std::integral_constant<int, T1 - 1>{});
        if (stamp) RB_STAMP(5);
        __syncthreads();                                         // B: t complete
        if (stamp) RB_STAMP(6);

        // ---------------- phase 2: out = (b2 + x) + conv2(t): the accumulators start as bias + residual ------------------------
        f32x4 c0[T2], c1[T2];
        {
            const f32x4 bv0 = *reinterpret_cast<const f32x4*>(smem + RB_BIAS + 128 + q * 16);
            const f32x4 bv1 = *reinterpret_cast<const f32x4*>(smem + RB_BIAS + 192 + q * 16);
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                if constexpr (HEAD != 0) {
                    c0[t] = bv0;                                     // [b0, b1, b2, 0] in lanes q = 0, zeros elsewhere (lo rows, pads)
                    c1[t] = bv1;
                } else {
                    const f16x4 x0 = xr0[t], x1 = xr1[t];
                    c0[t] = (f32x4){bv0[0] + (float)x0[0], bv0[1] + (float)x0[1], bv0[2] + (float)x0[2], bv0[3] + (float)x0[3]};
                    c1[t] = (f32x4){bv1[0] + (float)x1[0], bv1[1] + (float)x1[1], bv1[2] + (float)x1[2], bv1[3] + (float)x1[3]};
                }
            }
        }
        if constexpr (HEAD != 0) rb_kloop_head<T2>(c0, smem, RB_WB, la, pb2, delta6);
        else rb_kloop<T2, T2>(c0, c1, smem, RB_WB, la, pb2, delta6);
        if (stamp) RB_STAMP(7);
        if (has_next) {
            __syncthreads();                                     // C: every wave is done reading t
            x_park();
        }
        if (stamp) RB_STAMP(8);
        if constexpr (HEAD != 0) {
            // clamp( conv_last + bias + clamp01(bicubic(lr_centre)), 0, 1 ) -> planar fp32: after the fold lane (0, n) holds the
            // three channel sums of pixel n, lane (q, n), q < 3, takes channel q and evaluates ITS channel's bicubic sample
            const size_t plane_o = (size_t)p.h * p.w, plane_b = (size_t)p.bh * p.bw;
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                const f32x4 y = c0[t];
                const float s0 = rb_fold1(y[0]), s1 = rb_fold1(y[1]), s2 = rb_fold1(y[2]);
                const int srcl = lane & 15;
                const float v0 = __shfl(s0, srcl), v1 = __shfl(s1, srcl), v2 = __shfl(s2, srcl);
                const float v = q == 0 ? v0 : q == 1 ? v1 : v2;
                int lpe = lp;
                if (!interior) asm volatile("" : "+v"(lpe));
                const int oy = ty0 + oy0 + (t >> 1), ox = tx0 + (t & 1) * 16 + lpe;
                if (q < 3 && oy < p.h && ox < p.w) {
                    const float b = fminf(fmaxf(rv_bicubic_at(p.base_lr + q * plane_b, p.bh, p.bw, oy, ox, p.base_step, p.base_step), 0.0f), 1.0f);
                    rv_store_result(outp, q * plane_o + (size_t)oy * p.w + ox, fminf(fmaxf(v + b, 0.0f), 1.0f), p.out_fmt);
                }
            }
        } else if constexpr (STORE == 0) {
            unsigned char* ob = outp + ((long long)ty0 * p.w + tx0) * RB_PXB;
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                const f32x4 m = rb_fold_halves(c1[t]);
                const uint2 v0 = rb_pack(c0[t]), v1 = rb_pack(m);
                bool ok = true;
                if (!interior) {
                    int lpe = lp;
                    asm volatile("" : "+v"(lpe));
                    ok = ty0 + oy0 + (t >> 1) < p.h && tx0 + (t & 1) * 16 + lpe < p.w;
                }
                unsigned char* d = ob + (unsigned)((t >> 1) * rowb_g) + oo + (t & 1) * 16 * RB_PXB;
                if (ok) {
                    *reinterpret_cast<uint2*>(d) = v0;
                    if (q < 2) *reinterpret_cast<uint2*>(d + 32) = v1;
                }
            }
        } else {
            static_assert(T2 % 2 == 0, "the two pixel groups of an output row live in one wave");
            unsigned char* ob = outp + ((long long)ty0 * p.w + tx0) * RB_PXB;
            const int gsel = q & 1;                                  // this lane stores a pixel of the left (0) | right (1) group
#pragma unroll
            for (int tp = 0; tp < T2 / 2; ++tp) {
                const uint2 a0 = rb_pack(c0[2 * tp]), b0 = rb_pack(c0[2 * tp + 1]);
                const uint2 a1 = rb_pack(rb_fold_halves(c1[2 * tp])), b1 = rb_pack(rb_fold_halves(c1[2 * tp + 1]));
                // odd 16-lane rows of the left group's registers <-> even rows of the right group's: lane (q, pixel) ends up with
                // channels 8 (q >> 1) .. + 8 of its pixel in group q & 1 (and, for q < 2, channels 16-23 from the folded tile)
                const auto sx = __builtin_amdgcn_permlane16_swap(a0.x, b0.x, false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(a0.y, b0.y, false, false);
                const auto tx = __builtin_amdgcn_permlane16_swap(a1.x, b1.x, false, false);
                const auto ty_ = __builtin_amdgcn_permlane16_swap(a1.y, b1.y, false, false);
                const unsigned sx0 = sx[0], sx1 = sx[1], sy0 = sy[0], sy1 = sy[1], tx0_ = tx[0], tx1 = tx[1], ty0_ = ty_[0], ty1 = ty_[1];
                const u32x4 lo = {sx0, sy0, sx1, sy1};               // channels 8 (q >> 1) .. + 8
                const u32x4 hi = {tx0_, ty0_, tx1, ty1};              // q < 2: channels 16 .. 23
                bool ok = true;
                if (!interior) {
                    int lpe = lp;
                    asm volatile("" : "+v"(lpe));
                    ok = ty0 + oy0 + tp < p.h && tx0 + gsel * 16 + lpe < p.w;
                }
                unsigned char* d = ob + (unsigned)(tp * rowb_g) + (unsigned)(oy0 * rowb_g + (gsel * 16 + lp) * RB_PXB);
                if (ok) {
                    *reinterpret_cast<u32x4*>(d + (q >> 1) * 16) = lo;
                    if (q < 2) *reinterpret_cast<u32x4*>(d + 32) = hi;
                }
            }
        }
        if (stamp) RB_STAMP(9);
        if (has_next) __syncthreads();                           // D: next x tile visible
        if (stamp) RB_STAMP(10);
    }
    RB_STAMP(11);
#undef RB_STAMP
}

extern unsigned long long* g_rb_probe;             // runtime.hip: refvsr_set_probe
extern int g_rb_probe_iter;

static int g_rb24_store = REFVSR_RB24_STORE_DEFAULT;   // A/B knob (refvsr_set_resblock24_store): 0 | 1, see the kernel's STORE parameter
extern "C" int refvsr_set_resblock24_store(int mode) {
    if (mode < 0 || mode > 1) return 1;
    g_rb24_store = mode;
    return 0;
}
static int g_rb24_waves = 0;                 // A/B knob (refvsr_set_resblock24_waves): 0 = by map size; 4 | 8: 8 x 32 tiles; 16: 16 x 32
extern "C" int refvsr_set_resblock24_waves(int waves) {
    if (waves != 0 && waves != 4 && waves != 8 && waves != 16) return 1;
    g_rb24_waves = waves;
    return 0;
}

template <bool RELU, int NWV, bool PROBE = false, int TH = 8, int STORE = 0, int HEAD = 0>
static int launch_rb24(RB24Args& a, hipStream_t st) {
    constexpr int RB_LDS = rb_lds(TH);
    a.tiles_x = rv_cdiv(a.w, RB_TW);
    a.tpm = a.tiles_x * rv_cdiv(a.h, TH);
    a.n_tiles = a.tpm * (a.batch > 1 ? a.batch : 1);
    static bool attr_done[RV_MAX_DEVICES] = {};
    static int occ_dev[RV_MAX_DEVICES] = {};
    const int dev = rv_device();
    if (!attr_done[dev]) {
        RV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock24_kernel<RELU, NWV, PROBE, TH, STORE, HEAD>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, RB_LDS));
        int occ = 0;
        RV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, resblock24_kernel<RELU, NWV, PROBE, TH, STORE, HEAD>, NWV * 64, RB_LDS));
        occ_dev[dev] = occ < 1 ? 1 : occ;
        attr_done[dev] = true;
    }
    int cap = (rv_stream_cus(st) * occ_dev[dev]) & ~7;
    if (cap < 8) cap = 8;
    a.grid = a.n_tiles < cap ? a.n_tiles : cap;
    hipLaunchKernelGGL((resblock24_kernel<RELU, NWV, PROBE, TH, STORE, HEAD>), dim3(a.grid), dim3(NWV * 64), RB_LDS, st, a);
    RV_LAUNCH_CHECK();
    return 0;
}

// n fused blocks x <- x + conv2(act(conv1 x)) on `batch` 24-channel fp16 HWC maps of one geometry (batch = 1: the plain chain); block
// i's weights are the blob at blobs + i * blob_stride (refvsr_amd/packing.py:pack_resblock24).  n launches on the caller's stream, each
// over ALL maps (one weight fill per workgroup, batch x the tiles: an LR launch of RefVSR_small is one 8 x 32 tile per workgroup --
// 4 100 of its 13 000 cycles are the 43 KB fill -- two maps per launch are two tiles per fill); intermediates ping-pong between
// scratch0 / scratch1 ([batch] maps each, contiguous; blocks cannot run in place: neighbouring tiles read the input halo).
static int rb24_chain_impl(const void* const* src, int batch, int h, int w, int n, const void* blobs, size_t blob_stride,
                           float act_slope, void* scratch0, void* scratch1, void* const* out, void* stream) {
    RV_CHECK(src && out && blobs && h > 0 && w > 0 && n >= 1 && batch >= 1 && batch <= REFVSR_MAX_MAPS, "resblock24_chain: bad args");
    RV_CHECK(blob_stride >= (size_t)RB_BLOB && blob_stride % 16 == 0 && ((uintptr_t)blobs & 15) == 0,
             "resblock24_chain: blobs must be 16-byte aligned, stride >= %d", RB_BLOB);
    RV_CHECK(act_slope >= 0.f && act_slope <= 1.f, "resblock24_chain: activation slope must lie in [0, 1]");
    RV_CHECK(n == 1 || scratch0, "resblock24_chain: n >= 2 needs scratch0");
    RV_CHECK(n <= 2 || scratch1, "resblock24_chain: n >= 3 needs scratch1");
    const size_t mapb = (size_t)h * w * RB_PXB;
    for (int b = 0; b < batch; ++b) {
        RV_CHECK(src[b] && out[b], "resblock24_chain: null map pointer (map %d)", b);
        for (int c = 0; c < batch; ++c) {
            const unsigned char* s0 = scratch0 ? (const unsigned char*)scratch0 + c * mapb : nullptr;
            const unsigned char* s1 = scratch1 ? (const unsigned char*)scratch1 + c * mapb : nullptr;
            RV_CHECK(src[b] != out[c] && s0 != out[b] && s1 != out[b] && (n < 2 || s0 != src[b]) && (n < 3 || s1 != src[b]) &&
                     (c == b || out[b] != out[c]), "resblock24_chain: buffers must be distinct");
        }
    }
    RV_CHECK(n < 3 || scratch0 != scratch1, "resblock24_chain: buffers must be distinct");
    RV_CHECK((long long)h * w * RB_PXB < (1ll << 31), "resblock24_chain: map too large for 32-bit offsets");
    RV_CHECK(refvsr_init() == 0, "init failed");
    RB24Args a;
    memset(&a, 0, sizeof(a));
    a.h = h; a.w = w; a.act_slope = act_slope; a.batch = batch;
    hipStream_t st = (hipStream_t)stream;
    const unsigned char* cur[REFVSR_MAX_MAPS];
    for (int b = 0; b < batch; ++b) cur[b] = (const unsigned char*)src[b];
    for (int i = 0; i < n; ++i) {
        unsigned char* sc = (unsigned char*)((i & 1) ? scratch1 : scratch0);
        for (int b = 0; b < batch; ++b) {
            a.bsrc[b] = cur[b];
            a.bout[b] = (i == n - 1) ? (unsigned char*)out[b] : sc + b * mapb;
        }
        a.src = a.bsrc[0]; a.out = a.bout[0]; a.blob = (const unsigned char*)blobs + (size_t)i * blob_stride;
        int rc;
        a.probe = g_rb_probe; a.probe_iter = g_rb_probe_iter;
        // 16 x 32 tiles on sixteen waves (one workgroup per CU: half the weight fill per CU, 10 % less halo work in conv1, 17 % less
        // tile staging) pay on maps of many tiles per workgroup -- 540 x 960: 27.3 -> 26.3 us, 1080 x 1920: 100.1 -> 95.9 us; at
        // 270 x 480 (one tile per workgroup either way) the two shapes are equal (9.3 vs 9.4 us: the fill is latency, not bandwidth,
        // and sixteen waves wait longer at the barriers), below that the 8 x 32 tiles fill more CUs (135 x 240: 6.2 vs 8.2 us)
        const int nt8 = rv_cdiv(w, RB_TW) * rv_cdiv(h, 8) * batch;
        const int waves = g_rb24_waves ? g_rb24_waves : (nt8 >= 4 * rv_stream_cus(st) ? 16 : 8);
        if (g_rb_probe && act_slope == 0.f && waves == 8) rc = launch_rb24<true, 8, true>(a, st);          // tools/probe_resblock24.py
        else if (g_rb_probe && act_slope == 0.f && waves == 16) rc = launch_rb24<true, 16, true, 16>(a, st);
        else if (waves == 4) rc = act_slope == 0.f ? launch_rb24<true, 4>(a, st) : launch_rb24<false, 4>(a, st);
#define RB24_PICK(R_)                                                                                                              \
        (waves == 16 ? (g_rb24_store == 1 ? launch_rb24<R_, 16, false, 16, 1>(a, st) : launch_rb24<R_, 16, false, 16, 0>(a, st)) \
                     : (g_rb24_store == 1 ? launch_rb24<R_, 8, false, 8, 1>(a, st) : launch_rb24<R_, 8, false, 8, 0>(a, st)))
        else rc = act_slope == 0.f ? RB24_PICK(true) : RB24_PICK(false);
#undef RB24_PICK
        if (rc) return rc;
        for (int b = 0; b < batch; ++b) cur[b] = a.bout[b];
    }
    return 0;
}

extern "C" int refvsr_resblock24_chain(const void* src, int h, int w, int n, const void* blobs, size_t blob_stride,
                                       float act_slope, void* scratch0, void* scratch1, void* out, void* stream) {
    RV_CHECK(src && out, "resblock24_chain: bad args");
    return rb24_chain_impl(&src, 1, h, w, n, blobs, blob_stride, act_slope, scratch0, scratch1, &out, stream);
}

extern "C" int refvsr_resblock24_chain_batch(const void* const* src, int batch, int h, int w, int n, const void* blobs, size_t blob_stride,
                                             float act_slope, void* scratch0, void* scratch1, void* const* out, void* stream) {
    return rb24_chain_impl(src, batch, h, w, n, blobs, blob_stride, act_slope, scratch0, scratch1, out, stream);
}

// The last two convs of the upsampler in ONE launch (RefVSR.py:91-92,116-118,288,297, mid_channels = 24):
//   out = clamp( conv_last( lrelu_{act_slope}( conv_hr(src) ) ) + clamp01( F.interpolate(base_lr, bicubic) ), 0, 1 )   planar fp32 [3][h][w]
// src: fp16 HWC [h][w][24]; blob: REFVSR_RESBLOCK24_BLOB_BYTES in the block layout with conv1 = conv_hr and conv2's fragment slot
// (s, 0) = [rows 0-2: hi(W_last), rows 8-10: lo(W_last)], slots (s, 1), (s, 2) zero, b1 = conv_hr's bias, b2 = [b_last, 0 ...]
// (refvsr_amd/packing.py:pack_conv_hr_last).  The HR intermediate map (100 MB at 1080 x 1920) stays in LDS.
extern "C" int refvsr_conv_hr_last_fmt(const void* src, int h, int w, const void* blob, float act_slope, const float* base_lr, int bh, int bw,
                                       void* out, int out_fmt, void* stream);
extern "C" int refvsr_conv_hr_last(const void* src, int h, int w, const void* blob, float act_slope, const float* base_lr, int bh, int bw,
                                   float* out, void* stream) {
    return refvsr_conv_hr_last_fmt(src, h, w, blob, act_slope, base_lr, bh, bw, out, REFVSR_RESULT_F32, stream);
}
// ... with the result stored as fp32 | fp16 | uint8 = rint(255 v) (REFVSR_RESULT_*; ABI 14): host-side consumers quantise the frame to
// 8 bits anyway (evaluation/eval_qual_quan.py:117-119), a quarter of the bytes leaves the device
extern "C" int refvsr_conv_hr_last_fmt(const void* src, int h, int w, const void* blob, float act_slope, const float* base_lr, int bh, int bw,
                                       void* out, int out_fmt, void* stream) {
    RV_CHECK(src && out && blob && base_lr && h > 0 && w > 0, "conv_hr_last: bad args");
    RV_CHECK(out_fmt >= REFVSR_RESULT_F32 && out_fmt <= REFVSR_RESULT_U8, "conv_hr_last: unknown result format %d", out_fmt);
    RV_CHECK(((uintptr_t)blob & 15) == 0, "conv_hr_last: blob must be 16-byte aligned");
    RV_CHECK(act_slope > 0.f && act_slope <= 1.f, "conv_hr_last: activation slope must lie in (0, 1]");
    RV_CHECK(bh > 0 && bw > 0 && h % bh == 0 && w % bw == 0 && h / bh == w / bw, "conv_hr_last: base frame %dx%d does not divide the output %dx%d", bh, bw, h, w);
    RV_CHECK((long long)h * w * RB_PXB < (1ll << 31), "conv_hr_last: map too large for 32-bit offsets");
    RV_CHECK(refvsr_init() == 0, "init failed");
    RB24Args a;
    memset(&a, 0, sizeof(a));
    a.h = h; a.w = w; a.act_slope = act_slope;
    a.src = (const unsigned char*)src; a.out = (unsigned char*)out; a.blob = (const unsigned char*)blob;
    a.base_lr = base_lr; a.bh = bh; a.bw = bw; a.base_step = (float)bh / (float)h; a.out_fmt = out_fmt;
    hipStream_t st = (hipStream_t)stream;
    const int nt8 = rv_cdiv(w, RB_TW) * rv_cdiv(h, 8);
    const int waves = g_rb24_waves == 8 || g_rb24_waves == 16 ? g_rb24_waves : (nt8 >= 4 * rv_stream_cus(st) ? 16 : 8);
    return waves == 16 ? launch_rb24<false, 16, false, 16, 0, 1>(a, st) : launch_rb24<false, 8, false, 8, 0, 1>(a, st);
}

// K-block (K-step s, quarter q) of the blob's fragment order -> (ty, tx, cg) of the 3x3 x 24-channel window, or -1 for the
// zero block: the single source of truth for the host packer (refvsr_amd/packing.py checks itself against it).
extern "C" int refvsr_resblock24_kblock(int s, int q) {
    if (s < 0 || s >= RB_S || q < 0 || q > 3) return -2;
    int ty, u;
    if (s < 6) {
        const int perm[4] = {0, 2, 1, 3};
        ty = s >> 1;
        u = 4 * (s & 1) + perm[q];
    } else {
        if (q == 3) return -1;
        ty = q;
        u = 8;
    }
    return (ty << 16) | ((u / 3) << 8) | (u % 3);
}
