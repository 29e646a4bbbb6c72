// Fused residual block for the C = 48 maps of the RefVSR / RefVSR_MFID / RefVSR_MFID_8K family (mid_channels = 48, 30 blocks per
// propagation branch: configs/config_RefVSR_{L1,MFID,MFID_8K}.py):  out = x + conv2( act( conv1(x) ) ),  3x3, 48 -> 48, ONE launch
// per block.  Replaces the per-block body of ResidualBlockNoBN (mmedit/models/common/sr_backbone_utils.py:42-97; ReLU) and of
// ResBlock (models/archs/RefVSR_/common.py:25-39; LeakyReLU 0.2) for these models, which rounds 2-3 ran as two refvsr_conv48 launches
// with the intermediate map in HBM ("a fused block or a proof by measurement").
//
// The obstacle: one conv's hi + lo weights are 84 KB (14 K-steps x 6 fragments x 1 KiB), two sets + a tile
// do not fit 160 KB of LDS.  Here ONE set is resident at a time and the two sets swap per tile, by LDS-DMA, under the phases that
// do not read the weight region:
//
//   [W1 | x tile 12 x 36]   conv1 on the 10 x 34 halo region (K loop, W1)      residual x values -> registers
//   barrier A                every wave is done with W1 and the x tile
//   W2 -> weight region      global_load_lds, in flight ...
//   t = act(conv1) -> LDS    ... while the intermediate tile overwrites the x tile (inline-asm ds_write: hipcc orders every LDS
//                            access it knows about behind ALL outstanding LDS-DMA)
//   barrier B                t complete, W2 landed
//   conv2 on the 8 x 32 tile (K loop, W2); the residual is added in the epilogue (refvsr_conv48's summation order)
//   barrier C                every wave is done with W2 and t
//   W1 -> weight region      in flight while the next x tile (prefetched into registers during conv1) is parked and the
//   park next x, store out   output tile is stored
//   barrier D
//
// 168 KB of L2 -> LDS weight traffic per 8 x 32 tile instead of a 2 x 96-byte-per-pixel round trip of the intermediate map
// through HBM and a second launch.  Eight waves (two per SIMD, <= 256 VGPRs: two fragment sets in the K loop), one workgroup per CU
// (135 KB of LDS), persistent over the tiles (XCD-aware contiguous ranges, common.h).  K order and fragment layout are conv24.hip's
// COUT = 48 / NCG = 6 plan on a 36-pixel staged row (conv24_plan.h), so a block's parameter blob is the two refvsr_conv48 blobs'
// fragment parts back to back + the two bias vectors:
//   [W1: 14 x 6 x 1 KiB][W2: same][b1: 64 floats, 48.. = 0][b2: 64 floats]        (refvsr_amd/packing.py:pack_resblock48)
#include "common.h"
#include "conv24_plan.h"

namespace {
constexpr int R48_NCG = 6, R48_PS = 7, R48_PXB = R48_PS * 16;       // 112 bytes per staged pixel (six channel groups + one pad slot)
constexpr int R48_GPX = 96;                                         // bytes per pixel of the HWC maps in memory (48 halfs)
constexpr int R48_TH = 8, R48_TW = 32;
constexpr int R48_XH = R48_TH + 4, R48_XW = R48_TW + 4;             // x tile 12 x 36
constexpr int R48_IH = R48_TH + 2, R48_IW = R48_TW + 2;             // intermediate 10 x 34
constexpr int R48_ROWB = R48_XW * R48_PXB;                          // 4032
constexpr int R48_S = c24_steps(R48_NCG), R48_NF = 6, R48_NM = 3;   // 14 K-steps, six fragments ([hi | lo] x 3 sixteen-row tiles)
constexpr int R48_WB = R48_S * R48_NF * 1024;                       // 86016 bytes of fragments per conv
constexpr int R48_BIAS = R48_WB;                                    // LDS: [weights][b1 64 f][b2 64 f][x tile]
constexpr int R48_XT = R48_BIAS + 512;
constexpr int R48_XBYTES = R48_XH * R48_XW * R48_PXB;               // 48384
constexpr int R48_LDS = R48_XT + R48_XBYTES;                        // 134912
constexpr int R48_BLOB = 2 * R48_WB + 512;                          // 172544
constexpr int R48_NWV = 8, R48_NT = R48_NWV * 64;
constexpr int R48_NI = R48_IH * R48_IW;                             // 340 intermediate pixels
constexpr int R48_G1 = (R48_NI + 15) / 16;                          // 22 sixteen-pixel groups
constexpr int R48_T1 = (R48_G1 + R48_NWV - 1) / R48_NWV;            // 3
constexpr int R48_T1REM = R48_G1 % R48_NWV;                         // 6: waves below it have T1 groups, the others T1 - 1
constexpr int R48_T2 = 2 * R48_TH / R48_NWV;                        // 2
constexpr int R48_NCH = R48_XH * R48_XW * R48_NCG;                  // 2592 sixteen-byte chunks of the x tile
constexpr int R48_KCH = (R48_NCH + R48_NT - 1) / R48_NT;            // 6 per thread
constexpr int R48_NPAT = c24_npat(R48_NCG);
static_assert(c24_plan_ok(R48_NCG, R48_XW), "NCG = 6 K plan on the 36-pixel row");
static_assert(R48_BLOB == REFVSR_RESBLOCK48_BLOB_BYTES, "blob size is part of the C-ABI");
static_assert(R48_LDS <= 160 * 1024, "LDS budget");

typedef unsigned int r48_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int r48_u32x4 __attribute__((ext_vector_type(4)));
}  // namespace

struct RB48Args {
    const unsigned char* src; unsigned char* out; const unsigned char* blob;
    int h, w, tiles_x, n_tiles, grid;
    float act_slope;
    unsigned long long* probe;           // PROBE kernel: per-workgroup s_memtime stamps (refvsr_set_probe), 12 per workgroup
    int probe_iter;                      // which tile iteration of the workgroup is stamped
    // Multi-map launches (refvsr_resblock48_chain_batch): batch > 1 maps of one geometry share the launch -- its fixed cost, the first
    // weight fill and the tail; the two weight sets still swap per tile.  Flat tile index t = b * tpm + (tile of map b), map b reads
    // bsrc[b] and writes bout[b].  batch <= 1: src / out above.
    int batch, tpm;
    const unsigned char* bsrc[REFVSR_MAX_MAPS]; unsigned char* bout[REFVSR_MAX_MAPS];
};

// K loop of one conv: T pixel groups of this wave; fragments at LDS offset 0, B windows at pb[t] + pd[pattern] + immediate.
// Two fragment sets: the reads of step s + 1 are issued above the MFMAs of step s.
template <int T, int TA>
__device__ __forceinline__ void r48_kloop(f32x4 (&acc)[R48_NM][TA], const unsigned char* smem, const int la, const int (&pb)[TA],
                                          const int (&pd)[R48_NPAT]) {
    static_assert(T <= TA, "group count");
    uint4 fa[2][R48_NF], fb[2][T];
    auto load = [&](auto sc, uint4 (&af)[R48_NF], uint4 (&bf)[T]) {
        constexpr int s = decltype(sc)::value;
        constexpr int pp = c24_pat(R48_NCG, s);
        constexpr int imm = c24_off(R48_NCG, s, 0, R48_XW);
#pragma unroll
        for (int f = 0; f < R48_NF; ++f) af[f] = *reinterpret_cast<const uint4*>(smem + (s * R48_NF + f) * 1024 + la);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if constexpr (pp == 0) bf[t] = *reinterpret_cast<const uint4*>(smem + pb[t] + imm);
            else bf[t] = *reinterpret_cast<const uint4*>(smem + pb[t] + pd[pp] + imm);
        }
    };
    auto mfma = [&](const uint4 (&af)[R48_NF], const uint4 (&bf)[T]) {     // all hi products, then all lo products
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int m = 0; m < R48_NM; ++m) {
                const f16x8 av = *reinterpret_cast<const f16x8*>(&af[2 * m + hl]);
#pragma unroll
                for (int t = 0; t < T; ++t)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, *reinterpret_cast<const f16x8*>(&bf[t]), acc[m][t], 0, 0, 0);
            }
    };
    load(std::integral_constant<int, 0>{}, fa[0], fb[0]);
    c24_static_for([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + 1 < R48_S) load(std::integral_constant<int, s + 1>{}, fa[(s + 1) & 1], fb[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        mfma(fa[s & 1], fb[s & 1]);
    }, std::make_integer_sequence<int, R48_S>{});
}

template <bool RELU>
__device__ __forceinline__ r48_u32x2 r48_act_pack(const f32x4 y, const float slope) {
    union { f16x2 h; unsigned u; } a, b;
    if constexpr (RELU) {
        const f16x2 z = {(f16)0.f, (f16)0.f};
        a.h = __builtin_elementwise_max((f16x2){(f16)y[0], (f16)y[1]}, z);
        b.h = __builtin_elementwise_max((f16x2){(f16)y[2], (f16)y[3]}, z);
    } else {
        a.h = (f16x2){(f16)fmaxf(y[0], y[0] * slope), (f16)fmaxf(y[1], y[1] * slope)};
        b.h = (f16x2){(f16)fmaxf(y[2], y[2] * slope), (f16)fmaxf(y[3], y[3] * slope)};
    }
    return (r48_u32x2){a.u, b.u};
}

// PROBE (tools/probe_resblock48.py): stamps 0 entry; of tile `probe_iter`: 1 tile start (iteration 0: the first barrier -- W1, biases
// and the first x tile have landed), 2 conv1 K loop done, 3 barrier A, 4 t -> LDS issued and drained, 5 barrier B (W2 landed), 6 conv2 K loop done, 7 barrier C,
// 8 W1 DMA issued + next x tile parked, 9 output stores issued, 10 barrier D; 11 exit ([7]/[8]/[10] = the previous stamp when
// there is no next tile)
template <bool RELU, bool PROBE = false>
__global__ __launch_bounds__(R48_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void resblock48_kernel(RB48Args p) {
#define R48_STAMP(i) do { if constexpr (PROBE) { if (p.probe && threadIdx.x == 0) p.probe[blockIdx.x * 12 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
    R48_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("" :: "s"(p.src), "s"(p.out), "s"(p.blob), "s"(p.h), "s"(p.w), "s"(p.tiles_x), "s"(p.n_tiles), "s"(p.grid),
                 "s"(p.act_slope));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // weight set `which` (0 | 1) of the blob -> the weight region, 1 KiB per wave instruction
    auto w_dma = [&](const int which) {
        constexpr int NPC = R48_WB / 1024;                           // 84 pieces
        const unsigned char* g = p.blob + (size_t)which * R48_WB + lane * 16;
#pragma unroll
        for (int j = 0; j < (NPC + R48_NWV - 1) / R48_NWV; ++j) {
            const int c = wave + j * R48_NWV;
            if (c < NPC)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + c * 1024),
                                                 (__attribute__((address_space(3))) void*)(smem + c * 1024), 16, 0, 0);
        }
    };
    w_dma(0);
    if (wave == R48_NWV - 1 && lane < 32)                            // b1, b2: 512 bytes
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.blob + 2 * R48_WB + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + R48_BIAS), 16, 0, 0);

    // ---- x-tile chunks of this thread: i = tid + k NT = pixel * 6 + cg.  Their offsets are recomputed per tile from an opaque copy
    //      of the thread id (a dozen integer instructions per chunk): kept in registers across the K loops they cost 12 of the 256
    //      VGPRs a wave has at two waves per SIMD, and the kernel spilled
    const int rowb_g = p.w * R48_GPX;
    uint4 xv[R48_KCH];
    auto x_fetch = [&](const int tf) {
        int t = tf;
        const unsigned char* srcp = p.src;
        if (p.batch > 1) {                                           // flat tile index -> (map, tile of the map); uniform
            const int bm = (int)((unsigned)tf / (unsigned)p.tpm);
            t = tf - bm * p.tpm;
            srcp = p.bsrc[bm];
        }
        const int tyi = t / p.tiles_x;
        const int ty0 = tyi * R48_TH, tx0 = (t - tyi * p.tiles_x) * R48_TW;
        const bool interior = ty0 >= 2 && ty0 + R48_TH + 2 <= p.h && tx0 >= 2 && tx0 + R48_TW + 2 <= p.w;
        int tide = tid;
        asm volatile("" : "+v"(tide));
        if (interior) {
            const unsigned char* b = srcp + ((long long)(ty0 - 2) * p.w + (tx0 - 2)) * R48_GPX;
#pragma unroll
            for (int k = 0; k < R48_KCH; ++k) {
                const int i = min(tide + k * R48_NT, R48_NCH - 1);
                const int px = i / R48_NCG, cg = i - px * R48_NCG;
                const int r = px / R48_XW, c = px - r * R48_XW;
                xv[k] = *reinterpret_cast<const uint4*>(b + (unsigned)(r * rowb_g + c * R48_GPX + cg * 16));
            }
        } else {
#pragma unroll
            for (int k = 0; k < R48_KCH; ++k) {
                const int i = min(tide + k * R48_NT, R48_NCH - 1);
                const int px = i / R48_NCG, cg = i - px * R48_NCG;
                const int r = px / R48_XW, c = px - r * R48_XW;
                const int iy = ty0 - 2 + r, ix = tx0 - 2 + c;
                const bool ok = (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w;
                const unsigned off = (unsigned)((min(max(iy, 0), p.h - 1) * p.w + min(max(ix, 0), p.w - 1)) * R48_GPX + cg * 16);
                uint4 v = *reinterpret_cast<const uint4*>(srcp + off);           // clamped address, masked value (32-bit offsets: host check)
                const unsigned keep = ok ? 0xffffffffu : 0u;
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                xv[k] = v;
            }
        }
    };
    // registers -> LDS with inline-asm stores (a weight DMA may be in flight: see the header)
    auto x_park = [&]() {
        int tide = tid;
        asm volatile("" : "+v"(tide));
#pragma unroll
        for (int k = 0; k < R48_KCH; ++k) {
            const int i = tide + k * R48_NT;
            if (k * R48_NT + R48_NT <= R48_NCH || i < R48_NCH) {
                const int px = i / R48_NCG, cg = i - px * R48_NCG;
                const r48_u32x4 v = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
                const unsigned ad = lds0 + (unsigned)(R48_XT + px * R48_PXB + cg * 16);
                asm volatile("ds_write_b128 %0, %1" :: "v"(ad), "v"(v) : "memory");
            }
        }
    };

    int tl, k_hi;
    rv_tile_range(p.n_tiles, p.grid, tl, k_hi);
    if (tl < k_hi) x_fetch(tl);
    __builtin_amdgcn_sched_barrier(0);

    // ---- per-lane constants
    const int q = lane >> 4;
    const int lp = rv_pix16(lane & 15);
    const int la = lane * 16;
    auto sel4 = [&](const int v0, const int v1, const int v2, const int v3) { return q == 0 ? v0 : q == 1 ? v1 : q == 2 ? v2 : v3; };
    const int pat0 = sel4(c24_off(R48_NCG, 0, 0, R48_XW), c24_off(R48_NCG, 0, 1, R48_XW), c24_off(R48_NCG, 0, 2, R48_XW), c24_off(R48_NCG, 0, 3, R48_XW)) -
                     c24_off(R48_NCG, 0, 0, R48_XW);
    int pd[R48_NPAT];                                                // pattern p relative to pattern 0, per lane
#pragma unroll
    for (int pp = 0; pp < R48_NPAT; ++pp) {
        const int s = c24_pat_step(R48_NCG, pp);
        pd[pp] = (sel4(c24_off(R48_NCG, s, 0, R48_XW), c24_off(R48_NCG, s, 1, R48_XW), c24_off(R48_NCG, s, 2, R48_XW), c24_off(R48_NCG, s, 3, R48_XW)) -
                  c24_off(R48_NCG, s, 0, R48_XW)) - pat0;
    }
    const bool full1 = wave < R48_T1REM;
    const int g1 = full1 ? wave * R48_T1 : R48_T1REM * R48_T1 + (wave - R48_T1REM) * (R48_T1 - 1);
    int wo1[R48_T1], pb1[R48_T1];                                    // phase 1: window origin of each group's pixel (+ lane pattern)
#pragma unroll
    for (int t = 0; t < R48_T1; ++t) {
        const int pix = min((g1 + t) * 16 + lp, R48_NI - 1);         // lanes past the region repeat its last pixel
        const int r = pix / R48_IW;
        wo1[t] = R48_XT + r * R48_ROWB + (pix - r * R48_IW) * R48_PXB;
        pb1[t] = wo1[t] + pat0;
    }
    const int oy0 = wave;                                            // T2 = 2: the two 16-pixel groups of output row `wave`
    int wo2[R48_T2], pb2[R48_T2];
#pragma unroll
    for (int t = 0; t < R48_T2; ++t) {
        wo2[t] = R48_XT + (oy0 + 1) * R48_ROWB + (t * 16 + lp + 1) * R48_PXB;
        pb2[t] = wo2[t] + pat0;
    }
    const int cen = R48_ROWB + R48_PXB + q * 8;                      // channels 4 q .. of a window's centre pixel (+ 32 m per tile)

    if (tl < k_hi) x_park();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                                 // W1, biases, first x tile

    for (int iter = 0; tl < k_hi; ++tl, ++iter) {
        const bool has_next = tl + 1 < k_hi;
        const bool stamp = PROBE && iter == p.probe_iter;
        if (stamp) R48_STAMP(1);
        int tm = tl;
        unsigned char* outp = p.out;
        if (p.batch > 1) {
            const int bm = (int)((unsigned)tl / (unsigned)p.tpm);
            tm = tl - bm * p.tpm;
            outp = p.bout[bm];
        }
        const int tyi = tm / p.tiles_x;
        const int ty0 = tyi * R48_TH, tx0 = (tm - tyi * p.tiles_x) * R48_TW;
        const bool interior = ty0 >= 2 && ty0 + R48_TH + 2 <= p.h && tx0 >= 2 && tx0 + R48_TW + 2 <= p.w;

        // ---------------- phase 1: acc = b1 + conv1(x) on the halo region ----------------------------------------------------------
        f32x4 a1[R48_NM][R48_T1];
#pragma unroll
        for (int m = 0; m < R48_NM; ++m) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + R48_BIAS + m * 64 + q * 16);
#pragma unroll
            for (int t = 0; t < R48_T1; ++t) a1[m][t] = bv;
        }
        if (full1) r48_kloop<R48_T1, R48_T1>(a1, smem, la, pb1, pd);
        else r48_kloop<R48_T1 - 1, R48_T1>(a1, smem, la, pb1, pd);
        if (stamp) R48_STAMP(2);
        if (has_next) x_fetch(tl + 1);                               // next tile: in flight until it is parked after conv2
        // residual x values of this lane's outputs (the x tile is about to be overwritten by t)
        f16x4 xr[R48_NM][R48_T2];
#pragma unroll
        for (int t = 0; t < R48_T2; ++t)
#pragma unroll
            for (int m = 0; m < R48_NM; ++m) xr[m][t] = *reinterpret_cast<const f16x4*>(smem + wo2[t] + cen + 32 * m);
        __syncthreads();                                             // A: every wave is done with the x tile and W1
        if (stamp) R48_STAMP(3);
        w_dma(1);                                                    // W2 in flight under the t epilogue
        {
            auto epi1 = [&](auto tc) {
                constexpr int T = decltype(tc)::value;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    unsigned keep = 0xffffffffu;
                    if (!interior) {
                        int lpe = lp;
                        asm volatile("" : "+v"(lpe));
                        const int pix = min((g1 + t) * 16 + lpe, R48_NI - 1);
                        const int r = pix / R48_IW;
                        const int iy = ty0 - 1 + r, ix = tx0 - 1 + (pix - r * R48_IW);
                        keep = ((unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w) ? 0xffffffffu : 0u;
                    }
#pragma unroll
                    for (int m = 0; m < R48_NM; ++m) {
                        r48_u32x2 v = r48_act_pack<RELU>(a1[m][t], p.act_slope);
                        v.x &= keep; v.y &= keep;
                        const unsigned ad = lds0 + (unsigned)(wo1[t] + cen + 32 * m);     // (a local: asm operands cannot name captures)
                        asm volatile("ds_write_b64 %0, %1" :: "v"(ad), "v"(v) : "memory");
                    }
                }
            };
            if (full1) epi1(std::integral_constant<int, R48_T1>{}); else epi1(std::integral_constant<int, R48_T1 - 1>{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the asm stores above are invisible to hipcc's waitcnt pass
        if (stamp) R48_STAMP(4);
        __syncthreads();                                             // B: t complete, W2 landed (the fence waits for the DMA)
        if (stamp) R48_STAMP(5);

        // ---------------- phase 2: out = (b2 + conv2(t)) + x: the summation order of refvsr_conv48 with a residual operand ---------
        f32x4 c2[R48_NM][R48_T2];
#pragma unroll
        for (int m = 0; m < R48_NM; ++m) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + R48_BIAS + 256 + m * 64 + q * 16);
#pragma unroll
            for (int t = 0; t < R48_T2; ++t) c2[m][t] = bv;
        }
        r48_kloop<R48_T2, R48_T2>(c2, smem, la, pb2, pd);
        if (stamp) R48_STAMP(6);
        if (has_next) {
            __syncthreads();                                         // C: every wave is done with t and W2
            if (stamp) R48_STAMP(7);
            w_dma(0);                                                // W1 for the next tile, in flight under the park and the stores
            x_park();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (stamp) R48_STAMP(8);
        } else if (stamp) {
            R48_STAMP(7);
            R48_STAMP(8);
        }
        {
            unsigned char* ob = outp + ((long long)ty0 * p.w + tx0) * R48_GPX;
#pragma unroll
            for (int t = 0; t < R48_T2; ++t) {
                bool ok = true;
                if (!interior) {
                    int lpe = lp;
                    asm volatile("" : "+v"(lpe));
                    ok = ty0 + oy0 < p.h && tx0 + t * 16 + lpe < p.w;
                }
                unsigned char* d = ob + (unsigned)(oy0 * rowb_g + (t * 16 + lp) * R48_GPX + q * 8);
#pragma unroll
                for (int m = 0; m < R48_NM; ++m) {
                    const f32x4 y = c2[m][t];
                    const f16x4 x0 = xr[m][t];
                    const f16x4 o = {(f16)(y[0] + (float)x0[0]), (f16)(y[1] + (float)x0[1]), (f16)(y[2] + (float)x0[2]), (f16)(y[3] + (float)x0[3])};
                    if (ok) *reinterpret_cast<f16x4*>(d + 32 * m) = o;
                }
            }
        }
        if (stamp) R48_STAMP(9);
        if (has_next) __syncthreads();                               // D: W1 landed, next x tile visible
        if (stamp) R48_STAMP(10);
    }
    R48_STAMP(11);
#undef R48_STAMP
}

extern unsigned long long* g_rb_probe;             // runtime.hip: refvsr_set_probe
extern int g_rb_probe_iter;

template <bool RELU, bool PROBE = false>
static int launch_rb48(RB48Args& a, hipStream_t st) {
    static bool attr_done[RV_MAX_DEVICES] = {};
    const int dev = rv_device();
    if (!attr_done[dev]) {
        RV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock48_kernel<RELU, PROBE>), hipFuncAttributeMaxDynamicSharedMemorySize, R48_LDS));
        attr_done[dev] = true;
    }
    a.tiles_x = rv_cdiv(a.w, R48_TW);
    a.tpm = a.tiles_x * rv_cdiv(a.h, R48_TH);
    a.n_tiles = a.tpm * (a.batch > 1 ? a.batch : 1);
    int cap = rv_stream_cus(st) & ~7;                                     // one 135 KB workgroup per CU
    if (cap < 8) cap = 8;
    a.grid = a.n_tiles < cap ? a.n_tiles : cap;
    hipLaunchKernelGGL((resblock48_kernel<RELU, PROBE>), dim3(a.grid), dim3(R48_NT), R48_LDS, st, a);
    RV_LAUNCH_CHECK();
    return 0;
}

// n fused blocks x <- x + conv2(act(conv1 x)) on `batch` 48-channel fp16 HWC maps of one geometry (batch = 1: the plain chain); block i's
// parameters are the blob at blobs + i * blob_stride (refvsr_amd/packing.py:pack_resblock48).  n launches on the caller's stream, each over
// ALL maps; intermediates ping-pong between scratch0 / scratch1 ([batch] maps each, contiguous) like refvsr_resblock24_chain.
static int rb48_chain_impl(const void* const* src, int batch, int h, int w, int n, const void* blobs, size_t blob_stride, float act_slope,
                           void* scratch0, void* scratch1, void* const* out, void* stream) {
    RV_CHECK(src && out && blobs && h > 0 && w > 0 && n >= 1 && batch >= 1 && batch <= REFVSR_MAX_MAPS, "resblock48_chain: bad args");
    RV_CHECK(blob_stride >= (size_t)R48_BLOB && blob_stride % 16 == 0 && ((uintptr_t)blobs & 15) == 0,
             "resblock48_chain: blobs must be 16-byte aligned, stride >= %d", R48_BLOB);
    RV_CHECK(act_slope >= 0.f && act_slope <= 1.f, "resblock48_chain: activation slope must lie in [0, 1]");
    RV_CHECK(n == 1 || scratch0, "resblock48_chain: n >= 2 needs scratch0");
    RV_CHECK(n <= 2 || scratch1, "resblock48_chain: n >= 3 needs scratch1");
    const size_t mapb = (size_t)h * w * R48_GPX;
    for (int b = 0; b < batch; ++b) {
        RV_CHECK(src[b] && out[b], "resblock48_chain: null map pointer (map %d)", b);
        for (int c = 0; c < batch; ++c) {
            const unsigned char* s0 = scratch0 ? (const unsigned char*)scratch0 + c * mapb : nullptr;
            const unsigned char* s1 = scratch1 ? (const unsigned char*)scratch1 + c * mapb : nullptr;
            RV_CHECK(src[b] != out[c] && s0 != out[b] && s1 != out[b] && (n < 2 || s0 != src[b]) && (n < 3 || s1 != src[b]) &&
                     (c == b || out[b] != out[c]), "resblock48_chain: buffers must be distinct");
        }
    }
    RV_CHECK(n < 3 || scratch0 != scratch1, "resblock48_chain: buffers must be distinct");
    RV_CHECK((long long)h * w * R48_GPX < (1ll << 31), "resblock48_chain: map too large for 32-bit offsets");
    RV_CHECK(refvsr_init() == 0, "init failed");
    RB48Args a;
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
        a.probe = g_rb_probe; a.probe_iter = g_rb_probe_iter;
        const int rc = (g_rb_probe && act_slope == 0.f) ? launch_rb48<true, true>(a, st)       // tools/probe_resblock48.py
                       : act_slope == 0.f ? launch_rb48<true>(a, st) : launch_rb48<false>(a, st);
        if (rc) return rc;
        for (int b = 0; b < batch; ++b) cur[b] = a.bout[b];
    }
    return 0;
}

extern "C" int refvsr_resblock48_chain(const void* src, int h, int w, int n, const void* blobs, size_t blob_stride, float act_slope,
                                       void* scratch0, void* scratch1, void* out, void* stream) {
    RV_CHECK(src && out, "resblock48_chain: bad args");
    return rb48_chain_impl(&src, 1, h, w, n, blobs, blob_stride, act_slope, scratch0, scratch1, &out, stream);
}

extern "C" int refvsr_resblock48_chain_batch(const void* const* src, int batch, int h, int w, int n, const void* blobs, size_t blob_stride,
                                             float act_slope, void* scratch0, void* scratch1, void* const* out, void* stream) {
    return rb48_chain_impl(src, batch, h, w, n, blobs, blob_stride, act_slope, scratch0, scratch1, out, stream);
}
