// Fused residual block on CDNA4 matrix cores:  out = x + conv2( act( conv1(x) ) ),  both 3x3, C -> C,
// stride 1, zero padding -- ResidualBlockNoBN (mmedit sr_backbone_utils.py:42-97, act = ReLU) and ResBlock
// (RefVSR_/common.py:25-39, act = LeakyReLU 0.2).  ~70% of all convolutions of the network sit in such
// pairs; as two launches each pair costs two launch boundaries, two input stagings and an HBM round trip
// of the intermediate map.  Here one workgroup owns a 16x32 output tile:
//
//   stage   x tile (20 x 36 px, zero padded)            global -> LDS   (coalesced 16-byte HWC loads)
//           both packed weight sets (hi + lo fp16)      global -> LDS
//   phase 1 t = act(conv1(x) + b1) on the 18 x 34 halo region (612 px = 39 sixteen-pixel MFMA tiles,
//           flattened; positions outside the frame are forced to 0 = conv2's zero padding)  -> LDS (fp16)
//   phase 2 out = x + conv2(t) + b2 on the 16 x 32 tile; the residual x is re-read from the LDS tile
//           -> global (8-byte HWC channel vectors)
//
// Same MFMA orientation / K walk as conv_mfma.hip (D[cout][pixel], K-block = (tap, 8-channel group),
// v_mfma_f32_16x16x32_f16, hi+lo weights).  8 waves per workgroup (2 per SIMD).
//
// Workgroups are persistent: at most one per CU, each walking tiles  blockIdx.x, +gridDim.x, ...  The weights
// are fetched once per workgroup; the x tile is double buffered and the NEXT tile is prefetched into registers
// while the current one is in the matrix pipe (at LR there is one tile per CU and the loop runs once; on the
// 2x maps four tiles per workgroup hide three of four staging latencies and three of four weight loads).
#include "common.h"

#define RB_TH 16
#define RB_TW 32
#define RB_XH (RB_TH + 4)
#define RB_XW (RB_TW + 4)
#define RB_IH (RB_TH + 2)
#define RB_IW (RB_TW + 2)
#define RB_NI (RB_IH * RB_IW)            // 612 intermediate pixels
#define RB_T1 ((RB_NI + 15) / 16)        // 39 phase-1 tiles
#define RB_T1W ((RB_T1 + 7) / 8)         // 5 per wave
#define RB_T2W (RB_TH * 2 / 8)           // 4 per wave

struct ResBlockArgs {
    const f16* src; f16* out;
    int c, ncg, ps, h, w;
    int G, S;
    float inv_ncg;
    const uint4* w1; const float* b1;
    const uint4* w2; const float* b2;
    float act_slope, post_slope;
    int tab_bytes, w_bytes, x_bytes;     // LDS carve
    int tiles_x, n_tiles;
};

// Software-pipelined K loop shared by both phases: two fragment sets, unrolled by two (see conv_mfma.hip).
template <int MT, int T>
__device__ __forceinline__ void rb_kloop(f32x4 (&acc)[MT][T], const unsigned char* wl, const int* tq,
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

#define RB_XCH_MAX 5                      // uint4 prefetch registers per thread for one x tile (20*36*ncg / 512, ncg <= 3)

template <int MT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void resblock_mfma_kernel(ResBlockArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* tab1 = reinterpret_cast<int*>(smem);                    // K-block -> offset in the x tile
    int* tab2 = tab1 + p.S * 4;                                   // K-block -> offset in the t tile
    unsigned char* wl1 = smem + p.tab_bytes;
    unsigned char* wl2 = wl1 + p.w_bytes;
    unsigned char* xt0 = wl2 + p.w_bytes;                         // 2 x [RB_XH][RB_XW][ps*16]
    unsigned char* tt = xt0 + 2 * p.x_bytes;                      // [RB_IH][RB_IW][ps*16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = lane >> 4;
    const int lr = rv_pix16(lane & 15);      // pixel (of a 16-pixel MFMA tile) held by this lane's column
    const int psb = p.ps * 16;

    for (int g = tid; g < p.S * 4; g += 512) {                    // K-slot -> offset (K order: common.h:rv_kslot)
        int o1, o2, slot;
        if (g < p.G) {
            const int tap = (int)(((float)g + 0.5f) * p.inv_ncg);
            const int cg = g - tap * p.ncg;
            const int ty = tap / 3;
            const int tx = tap - ty * 3;
            o1 = ((ty * RB_XW + tx) * p.ps + cg) * 16;
            o2 = ((ty * RB_IW + tx) * p.ps + cg) * 16;
            slot = rv_kslot(ty, tx, cg, 3, p.ncg);
        } else {                                                  // zero-weight blocks: offset of the partner's parity
            slot = rv_kpad_slot(g - p.G, 3, p.ncg);
            o1 = o2 = (slot < p.G) ? 0 : (p.ncg > 1 ? 16 : p.ps * 16);
        }
        tab1[slot] = o1;
        tab2[slot] = o2;
    }
    const int n16w = p.S * MT * 2 * 64;            // uint4 per weight set (<= 4 * 512 for the supported shapes)
    for (int i = tid; i < n16w; i += 512) reinterpret_cast<uint4*>(wl1)[i] = p.w1[i];
    // x-tile chunk bookkeeping: chunk idx -> (pixel, channel group); identical for every tile
    const int row_chunks = RB_XW * p.ncg;
    const int total = RB_XH * row_chunks;
    const float inv_rc = 1.0f / (float)row_chunks;
    int xr[RB_XCH_MAX], xc[RB_XCH_MAX], xoff[RB_XCH_MAX], xcg[RB_XCH_MAX];
#pragma unroll
    for (int k = 0; k < RB_XCH_MAX; ++k) {
        const int idx = min(tid + k * 512, total - 1);
        const int r = (int)(((float)idx + 0.5f) * inv_rc);
        const int i = idx - r * row_chunks;
        const int c = (int)(((float)i + 0.5f) * p.inv_ncg);
        xr[k] = r; xc[k] = c; xcg[k] = i - c * p.ncg;
        xoff[k] = (r * RB_XW + c) * psb + xcg[k] * 16;
    }
    uint4 xv[RB_XCH_MAX];
    auto x_fetch = [&](int tile) {                 // global -> registers (zero padded at the frame border)
        const int ty0 = (tile / p.tiles_x) * RB_TH, tx0 = (tile % p.tiles_x) * RB_TW;
#pragma unroll
        for (int k = 0; k < RB_XCH_MAX; ++k) {
            const int iy = ty0 - 2 + xr[k], ix = tx0 - 2 + xc[k];
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w)
                v = *reinterpret_cast<const uint4*>(p.src + ((size_t)iy * p.w + ix) * p.c + xcg[k] * 8);
            xv[k] = v;
        }
    };
    auto x_park = [&](unsigned char* xt) {         // registers -> LDS
#pragma unroll
        for (int k = 0; k < RB_XCH_MAX; ++k)
            if (tid + k * 512 < total) *reinterpret_cast<uint4*>(xt + xoff[k]) = xv[k];
    };

    int tile = blockIdx.x;
    if (tile < p.n_tiles) { x_fetch(tile); x_park(xt0); }
    // conv2's weights are not needed before phase 2 of the first tile: fetch them into (named) registers now and park
    // them in LDS behind phase 1, so only x + w1 sit on the start-up critical path
    uint4 v0, v1, v2, v3;
    {
        const int last = n16w - 1;
        v0 = p.w2[min(tid, last)];
        v1 = p.w2[min(tid + 512, last)];
        v2 = p.w2[min(tid + 1024, last)];
        v3 = p.w2[min(tid + 1536, last)];
    }
    bool w2_parked = false;
    float4 b1r[MT], b2r[MT];                        // biases of this lane's output channels: fetched once
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co0 = min(m * 16 + q * 4, p.c - 4);
        b1r[m] = *reinterpret_cast<const float4*>(p.b1 + co0);
        b2r[m] = *reinterpret_cast<const float4*>(p.b2 + co0);
    }
    __syncthreads();

    int cur = 0;
    for (; tile < p.n_tiles; tile += gridDim.x) {
        const int ty0 = (tile / p.tiles_x) * RB_TH, tx0 = (tile % p.tiles_x) * RB_TW;
        const unsigned char* xt = xt0 + (size_t)cur * p.x_bytes;
        const bool has_next = (tile + (int)gridDim.x < p.n_tiles);
        if (has_next) x_fetch(tile + gridDim.x);    // in flight during both phases
        asm volatile("" ::: "memory");

        // ---------------- phase 1: t = act(conv1(x) + b1) on the halo region -------------------------
        {
            int pb[RB_T1W];
#pragma unroll
            for (int t = 0; t < RB_T1W; ++t) {
                int pix = (wave * RB_T1W + t) * 16 + lr;
                pix = min(pix, RB_NI - 1);
                const int r = (int)(((float)pix + 0.5f) * (1.0f / (float)RB_IW));
                const int c = pix - r * RB_IW;
                pb[t] = (r * RB_XW + c) * psb;
            }
            f32x4 acc[MT][RB_T1W];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < RB_T1W; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            rb_kloop<MT, RB_T1W>(acc, wl1, tab1 + q, xt, pb, p.S, lane);
#pragma unroll
            for (int t = 0; t < RB_T1W; ++t) {
                const int tl = wave * RB_T1W + t;
                const int pix = tl * 16 + lr;
                if (tl >= RB_T1 || pix >= RB_NI) continue;
                const int r = (int)(((float)pix + 0.5f) * (1.0f / (float)RB_IW));
                const int c = pix - r * RB_IW;
                const int iy = ty0 - 1 + r, ix = tx0 - 1 + c;
                const bool inside = (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int co0 = m * 16 + q * 4;
                    if (co0 >= p.c) continue;
                    const float4 bv = b1r[m];
                    f16x4 o;
                    o[0] = (f16)(inside ? rv_lrelu(acc[m][t][0] + bv.x, p.act_slope) : 0.f);
                    o[1] = (f16)(inside ? rv_lrelu(acc[m][t][1] + bv.y, p.act_slope) : 0.f);
                    o[2] = (f16)(inside ? rv_lrelu(acc[m][t][2] + bv.z, p.act_slope) : 0.f);
                    o[3] = (f16)(inside ? rv_lrelu(acc[m][t][3] + bv.w, p.act_slope) : 0.f);
                    *reinterpret_cast<f16x4*>(tt + (size_t)pix * psb + co0 * 2) = o;
                }
            }
        }
        if (!w2_parked) {
            uint4* d = reinterpret_cast<uint4*>(wl2);
            if (tid < n16w) d[tid] = v0;
            if (tid + 512 < n16w) d[tid + 512] = v1;
            if (tid + 1024 < n16w) d[tid + 1024] = v2;
            if (tid + 1536 < n16w) d[tid + 1536] = v3;
            w2_parked = true;
        }
        // the next x tile (loads issued before phase 1) goes to the other LDS buffer here, not after the output stores:
        // waiting for it there also waited for every store of this tile (vmcnt is in order)
        if (has_next) x_park(xt0 + (size_t)(cur ^ 1) * p.x_bytes);
        __syncthreads();

        // ---------------- phase 2: out = x + conv2(t) + b2 ------------------------------------------------
        {
            int pb[RB_T2W];
#pragma unroll
            for (int t = 0; t < RB_T2W; ++t) {
                const int ti = wave * RB_T2W + t;
                const int row = ti >> 1;
                const int col = (ti & 1) * 16 + lr;
                pb[t] = (row * RB_IW + col) * psb;
            }
            f32x4 acc[MT][RB_T2W];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < RB_T2W; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            rb_kloop<MT, RB_T2W>(acc, wl2, tab2 + q, tt, pb, p.S, lane);
#pragma unroll
            for (int t = 0; t < RB_T2W; ++t) {
                const int ti = wave * RB_T2W + t;
                const int row = ti >> 1;
                const int col = (ti & 1) * 16 + lr;
                const int oy = ty0 + row, ox = tx0 + col;
                if (oy >= p.h || ox >= p.w) continue;
                const unsigned char* xr_ = xt + (size_t)((row + 2) * RB_XW + (col + 2)) * psb;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int co0 = m * 16 + q * 4;
                    if (co0 >= p.c) continue;
                    const float4 bv = b2r[m];
                    const f16x4 xv4 = *reinterpret_cast<const f16x4*>(xr_ + co0 * 2);
                    float y[4] = {acc[m][t][0] + bv.x + (float)xv4[0], acc[m][t][1] + bv.y + (float)xv4[1],
                                  acc[m][t][2] + bv.z + (float)xv4[2], acc[m][t][3] + bv.w + (float)xv4[3]};
                    if (p.post_slope != 1.0f) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[i] = rv_lrelu(y[i], p.post_slope);
                    }
                    f16x4 o = {(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
                    *reinterpret_cast<f16x4*>(p.out + ((size_t)oy * p.w + ox) * p.c + co0) = o;
                }
            }
        }
        __syncthreads();                   // t tile free for the next phase 1
        cur ^= 1;
    }
}

template <int MT>
static int launch_resblock(const ResBlockArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        RV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_mfma_kernel<MT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL((resblock_mfma_kernel<MT>), grid, dim3(512), lds, st, a);
    RV_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================================================
// Two chained residual blocks in ONE launch:  y = x + conv2(act(conv1(x)));  out = y + conv4(act(conv3(y))).
// On the LR maps a fused block is one 16x32 tile per CU and ~13 us of which ~3.5 us are MFMA time: launch, weight / tile
// fetch and the store tail dominate, and 96-128 such launches per frame sit on the serial chain of the propagation
// branches.  Chaining two blocks pays 1.19x the MFMAs (4-pixel halo recomputation) for half the launches and half the
// HBM round trips.  One workgroup = one 16x32 output tile (not persistent: meant for maps with <= ~1 tile per CU):
//
//   X [24][40]   x tile (halo 4, zero padded)                     -> later y (halo 2) IN PLACE at offset (2, 2)
//   T [22][38]   t1 = act(conv1 x) on 22x38  (phase 1)            -> later t2 = act(conv3 y) on 18x34  (phase 3)
//   phase 2: y  = post1(x + conv2 t1) on 20x36, 0 outside the frame (= conv3's zero padding)
//   phase 4: out = post2(y + conv4 t2) on 16x32 -> global
//   weights ping-pong through two LDS buffers (w1 -> w3, w2 -> w4), the next set is fetched into named registers
//   during the current phase.
// Bit-identical to two refvsr_resblock_mfma launches (same K order, same fp16 rounding points).
#define RC_XH 24
#define RC_XW 40
#define RC_TH 22
#define RC_TW 38
#define RC_XCH 6                          // uint4 per thread for the x tile (24*40*ncg / 512, ncg <= 3)

struct ResChainArgs {
    const f16* src; f16* out;
    int c, ncg, ps, h, w;
    int G, S;
    float inv_ncg;
    const uint4* wq[4]; const float* bq[4];
    float act_slope, post1, post2;
    int tab_bytes, w_bytes, x_bytes;
    int tiles_x;
};

// one conv phase over a flattened RH x RW pixel region: region pixel (r, c) reads its 3x3 window at source pixel
// (r + oy, c + ox) of an LDS map with `pitch` pixels per row; T 16-pixel MFMA tiles per wave
template <int MT, int T, typename Epi>
__device__ __forceinline__ void rc_phase(const unsigned char* src, const int pitch, const int oy, const int ox,
                                         const int RH, const int RW, const unsigned char* wl, const int* tq, const int S,
                                         const int psb, const int wave, const int lane, const int lp, Epi epi) {
    const int NP = RH * RW;
    const float inv_rw = 1.0f / (float)RW;
    int pb[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int pix = min((wave * T + t) * 16 + lp, NP - 1);
        const int r = (int)(((float)pix + 0.5f) * inv_rw);
        const int c = pix - r * RW;
        pb[t] = ((r + oy) * pitch + (c + ox)) * psb;
    }
    f32x4 acc[MT][T];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    rb_kloop<MT, T>(acc, wl, tq, src, pb, S, lane);
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int pix = (wave * T + t) * 16 + lp;
        if (pix >= NP) continue;
        const int r = (int)(((float)pix + 0.5f) * inv_rw);
        const int c = pix - r * RW;
#pragma unroll
        for (int m = 0; m < MT; ++m) epi(r, c, m, acc[m][t]);
    }
}

template <int MT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void resblock_chain2_kernel(ResChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* tabx = reinterpret_cast<int*>(smem);                     // K-slot -> offset in a pitch-RC_XW map
    int* tabt = tabx + p.S * 4;                                    // K-slot -> offset in a pitch-RC_TW map
    unsigned char* wa = smem + p.tab_bytes;                        // weights of phases 1, 3
    unsigned char* wb = wa + p.w_bytes;                            // weights of phases 2, 4
    unsigned char* X = wb + p.w_bytes;                             // [RC_XH][RC_XW][ps*16]
    unsigned char* Tm = X + p.x_bytes;                             // [RC_TH][RC_TW][ps*16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = lane >> 4;
    const int lp = rv_pix16(lane & 15);
    const int psb = p.ps * 16;
    const int ty0 = (blockIdx.x / p.tiles_x) * RB_TH, tx0 = (blockIdx.x % p.tiles_x) * RB_TW;

    for (int g = tid; g < p.S * 4; g += 512) {                    // K order: common.h:rv_kslot
        int o1, o2, slot;
        if (g < p.G) {
            const int tap = (int)(((float)g + 0.5f) * p.inv_ncg);
            const int cg = g - tap * p.ncg;
            const int ty = tap / 3;
            const int tx = tap - ty * 3;
            o1 = ((ty * RC_XW + tx) * p.ps + cg) * 16;
            o2 = ((ty * RC_TW + tx) * p.ps + cg) * 16;
            slot = rv_kslot(ty, tx, cg, 3, p.ncg);
        } else {
            slot = rv_kpad_slot(g - p.G, 3, p.ncg);
            o1 = o2 = (slot < p.G) ? 0 : (p.ncg > 1 ? 16 : p.ps * 16);
        }
        tabx[slot] = o1;
        tabt[slot] = o2;
    }
    const int n16w = p.S * MT * 2 * 64;                           // uint4 per weight set (<= 4 * 512)
    for (int i = tid; i < n16w; i += 512) reinterpret_cast<uint4*>(wa)[i] = p.wq[0][i];
    {                                                             // x tile, halo 4, zero padded at the frame border
        const int row_chunks = RC_XW * p.ncg;
        const int total = RC_XH * row_chunks;
        const float inv_rc = 1.0f / (float)row_chunks;
#pragma unroll
        for (int k = 0; k < RC_XCH; ++k) {
            const int idx = tid + k * 512;
            if (idx < total) {
                const int r = (int)(((float)idx + 0.5f) * inv_rc);
                const int i = idx - r * row_chunks;
                const int c = (int)(((float)i + 0.5f) * p.inv_ncg);
                const int cg = i - c * p.ncg;
                const int iy = ty0 - 4 + r, ix = tx0 - 4 + c;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w)
                    v = *reinterpret_cast<const uint4*>(p.src + ((size_t)iy * p.w + ix) * p.c + cg * 8);
                *reinterpret_cast<uint4*>(X + (size_t)(r * RC_XW + c) * psb + cg * 16) = v;
            }
        }
    }
    // the next weight set travels through four named registers while the current phase computes
    uint4 v0, v1, v2, v3;
    const int last = n16w - 1;
#define RC_W_FETCH(W) { v0 = (W)[min(tid, last)]; v1 = (W)[min(tid + 512, last)]; v2 = (W)[min(tid + 1024, last)]; v3 = (W)[min(tid + 1536, last)]; }
#define RC_W_PARK(D) { uint4* d_ = reinterpret_cast<uint4*>(D); if (tid < n16w) d_[tid] = v0; if (tid + 512 < n16w) d_[tid + 512] = v1; \
                       if (tid + 1024 < n16w) d_[tid + 1024] = v2; if (tid + 1536 < n16w) d_[tid + 1536] = v3; }
    RC_W_FETCH(p.wq[1])
    float4 br[4][MT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) br[j][m] = *reinterpret_cast<const float4*>(p.bq[j] + min(m * 16 + q * 4, p.c - 4));
    __syncthreads();

    auto act4 = [&](const f32x4 a, const float4 b, const float slope, const bool keep) -> f16x4 {
        f16x4 o;
        o[0] = (f16)(keep ? rv_lrelu(a[0] + b.x, slope) : 0.f);
        o[1] = (f16)(keep ? rv_lrelu(a[1] + b.y, slope) : 0.f);
        o[2] = (f16)(keep ? rv_lrelu(a[2] + b.z, slope) : 0.f);
        o[3] = (f16)(keep ? rv_lrelu(a[3] + b.w, slope) : 0.f);
        return o;
    };
    auto res4 = [&](const f32x4 a, const float4 b, const f16x4 x, const float post, const bool keep) -> f16x4 {
        float y[4] = {a[0] + b.x + (float)x[0], a[1] + b.y + (float)x[1], a[2] + b.z + (float)x[2], a[3] + b.w + (float)x[3]};
        if (post != 1.0f) {
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = rv_lrelu(y[i], post);
        }
        f16x4 o = {(f16)(keep ? y[0] : 0.f), (f16)(keep ? y[1] : 0.f), (f16)(keep ? y[2] : 0.f), (f16)(keep ? y[3] : 0.f)};
        return o;
    };

    // ---- phase 1: t1 = act(conv1(x) + b1) on 22 x 38, origin (ty0 - 3, tx0 - 3) ---------------------------------------
    rc_phase<MT, 7>(X, RC_XW, 0, 0, RC_TH, RC_TW, wa, tabx + q, p.S, psb, wave, lane, lp,
                    [&](const int r, const int c, const int m, const f32x4 a) {
                        const int co0 = m * 16 + q * 4;
                        if (co0 >= p.c) return;
                        const int iy = ty0 - 3 + r, ix = tx0 - 3 + c;
                        const bool inside = iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
                        *reinterpret_cast<f16x4*>(Tm + (size_t)(r * RC_TW + c) * psb + co0 * 2) = act4(a, br[0][m], p.act_slope, inside);
                    });
    RC_W_PARK(wb)                                                 // w2 (wb was never used before)
    RC_W_FETCH(p.wq[2])
    __syncthreads();
    // ---- phase 2: y = post1(x + conv2(t1) + b2) on 20 x 36, origin (ty0 - 2, tx0 - 2); in place over x at (+2, +2) -------
    rc_phase<MT, 6>(Tm, RC_TW, 0, 0, 20, 36, wb, tabt + q, p.S, psb, wave, lane, lp,
                    [&](const int r, const int c, const int m, const f32x4 a) {
                        const int co0 = m * 16 + q * 4;
                        if (co0 >= p.c) return;
                        const int iy = ty0 - 2 + r, ix = tx0 - 2 + c;
                        const bool inside = iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
                        unsigned char* xp = X + (size_t)((r + 2) * RC_XW + (c + 2)) * psb + co0 * 2;
                        const f16x4 xv = *reinterpret_cast<const f16x4*>(xp);
                        *reinterpret_cast<f16x4*>(xp) = res4(a, br[1][m], xv, p.post1, inside);
                    });
    RC_W_PARK(wa)                                                 // w3 (every wave left phase 1, the last reader of wa,
    RC_W_FETCH(p.wq[3])                                           //     at the barrier above)
    __syncthreads();                                              // y complete, w3 visible
    // ---- phase 3: t2 = act(conv3(y) + b3) on 18 x 34, origin (ty0 - 1, tx0 - 1); window origin X(r + 2, c + 2) ---------
    rc_phase<MT, 5>(X, RC_XW, 2, 2, 18, 34, wa, tabx + q, p.S, psb, wave, lane, lp,
                    [&](const int r, const int c, const int m, const f32x4 a) {
                        const int co0 = m * 16 + q * 4;
                        if (co0 >= p.c) return;
                        const int iy = ty0 - 1 + r, ix = tx0 - 1 + c;
                        const bool inside = iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
                        *reinterpret_cast<f16x4*>(Tm + (size_t)(r * RC_TW + c) * psb + co0 * 2) = act4(a, br[2][m], p.act_slope, inside);
                    });
    RC_W_PARK(wb)                                                 // w4 (every wave left phase 2 at the barrier above)
    __syncthreads();                                              // t2 complete, w4 visible
    // ---- phase 4: out = post2(y + conv4(t2) + b4) on 16 x 32 -> global ------------------------------------------------
    rc_phase<MT, 4>(Tm, RC_TW, 0, 0, RB_TH, RB_TW, wb, tabt + q, p.S, psb, wave, lane, lp,
                    [&](const int r, const int c, const int m, const f32x4 a) {
                        const int co0 = m * 16 + q * 4;
                        const int oy = ty0 + r, ox = tx0 + c;
                        if (co0 >= p.c || oy >= p.h || ox >= p.w) return;
                        const f16x4 yv = *reinterpret_cast<const f16x4*>(X + (size_t)((r + 4) * RC_XW + (c + 4)) * psb + co0 * 2);
                        *reinterpret_cast<f16x4*>(p.out + ((size_t)oy * p.w + ox) * p.c + co0) = res4(a, br[3][m], yv, p.post2, true);
                    });
#undef RC_W_FETCH
#undef RC_W_PARK
}

static size_t rc_lds_bytes(int c) {
    const int ncg = c / 8, ps = ncg | 1;
    const int S = rv_ksteps(3, ncg);
    const int MT = (c + 15) / 16;
    return (size_t)((S * 4 * 2 * 4 + 15) / 16 * 16) + 2 * (size_t)S * MT * 2 * 1024 +
           (size_t)RC_XH * RC_XW * ps * 16 + (size_t)RC_TH * RC_TW * ps * 16;
}

extern "C" int refvsr_resblock2_fits(int c) {
    if (c <= 0 || c % 8 != 0) return 0;
    const int ncg = c / 8;
    const int S = rv_ksteps(3, ncg);
    const int MT = (c + 15) / 16;
    if (MT > 2) return 0;
    if (RC_XH * RC_XW * ncg > RC_XCH * 512) return 0;             // x-tile staging slots
    if (S * MT * 2 * 64 > 4 * 512) return 0;                      // weight prefetch registers
    return rc_lds_bytes(c) <= 160 * 1024 ? 1 : 0;
}

template <int MT>
static int launch_chain2(const ResChainArgs& a, int n_tiles, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        RV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_chain2_kernel<MT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL((resblock_chain2_kernel<MT>), dim3(n_tiles), dim3(512), lds, st, a);
    RV_LAUNCH_CHECK();
    return 0;
}

extern "C" int refvsr_resblock2_mfma(const void* src, int c, int h, int w, const void* const* wq, const float* const* bq,
                                     int ksteps, float act_slope, float post1, float post2, void* out, void* stream) {
    RV_CHECK(src && out && wq && bq && h > 0 && w > 0, "resblock2: bad args");
    for (int j = 0; j < 4; ++j) RV_CHECK(wq[j] && bq[j], "resblock2: null weights / bias %d", j);
    RV_CHECK(src != out, "resblock2: in-place operation is not supported (neighbouring tiles read the input halo)");
    RV_CHECK(refvsr_resblock2_fits(c), "resblock2: channel count %d not supported by the chained kernel", c);
    RV_CHECK(refvsr_init() == 0, "init failed");
    ResChainArgs a;
    memset(&a, 0, sizeof(a));
    a.src = (const f16*)src; a.out = (f16*)out;
    a.c = c; a.ncg = c / 8; a.ps = a.ncg | 1; a.h = h; a.w = w;
    a.G = 9 * a.ncg; a.S = rv_ksteps(3, a.ncg);
    RV_CHECK(a.S == ksteps, "resblock2: ksteps mismatch (%d vs %d)", ksteps, a.S);
    a.inv_ncg = 1.0f / (float)a.ncg;
    for (int j = 0; j < 4; ++j) { a.wq[j] = (const uint4*)wq[j]; a.bq[j] = bq[j]; }
    a.act_slope = act_slope; a.post1 = post1; a.post2 = post2;
    const int MT = (c + 15) / 16;
    a.tab_bytes = (a.S * 4 * 2 * 4 + 15) / 16 * 16;
    a.w_bytes = a.S * MT * 2 * 1024;
    a.x_bytes = RC_XH * RC_XW * a.ps * 16;
    a.tiles_x = rv_cdiv(w, RB_TW);
    const int n_tiles = a.tiles_x * rv_cdiv(h, RB_TH);
    const size_t lds = rc_lds_bytes(c);
    if (MT == 1) return launch_chain2<1>(a, n_tiles, lds, (hipStream_t)stream);
    return launch_chain2<2>(a, n_tiles, lds, (hipStream_t)stream);
}

extern "C" int refvsr_resblock_fits(int c) {
    if (c <= 0 || c % 8 != 0) return 0;
    const int ncg = c / 8, ps = ncg | 1;
    const int S = rv_ksteps(3, ncg);
    const int MT = (c + 15) / 16;
    if (MT > 2) return 0;
    if (RB_XH * RB_XW * ncg > RB_XCH_MAX * 512) return 0;        // x-tile prefetch registers
    if (S * MT * 2 * 64 > 4 * 512) return 0;                     // conv2 weight prefetch registers
    const size_t lds = (size_t)((S * 4 * 2 * 4 + 15) / 16 * 16) + 2 * (size_t)S * MT * 2 * 1024 +
                       2 * (size_t)RB_XH * RB_XW * ps * 16 + (size_t)RB_IH * RB_IW * ps * 16;
    return lds <= 160 * 1024 ? 1 : 0;
}

extern "C" int refvsr_resblock_mfma(const void* src, int c, int h, int w, const void* w1, const float* b1,
                                    const void* w2, const float* b2, int ksteps, float act_slope, float post_slope,
                                    void* out, void* stream) {
    RV_CHECK(src && out && w1 && w2 && b1 && b2 && h > 0 && w > 0, "resblock: bad args");
    RV_CHECK(src != out, "resblock: in-place operation is not supported (neighbouring tiles read the input halo)");
    RV_CHECK(refvsr_resblock_fits(c), "resblock: channel count %d not supported by the fused kernel", c);
    RV_CHECK(refvsr_init() == 0, "init failed");
    ResBlockArgs a;
    memset(&a, 0, sizeof(a));
    a.src = (const f16*)src; a.out = (f16*)out;
    a.c = c; a.ncg = c / 8; a.ps = a.ncg | 1; a.h = h; a.w = w;
    a.G = 9 * a.ncg; a.S = rv_ksteps(3, a.ncg);
    RV_CHECK(a.S == ksteps, "resblock: ksteps mismatch (%d vs %d)", ksteps, a.S);
    a.inv_ncg = 1.0f / (float)a.ncg;
    a.w1 = (const uint4*)w1; a.b1 = b1; a.w2 = (const uint4*)w2; a.b2 = b2;
    a.act_slope = act_slope; a.post_slope = post_slope;
    const int MT = (c + 15) / 16;
    a.tab_bytes = (a.S * 4 * 2 * 4 + 15) / 16 * 16;
    a.w_bytes = a.S * MT * 2 * 1024;
    a.x_bytes = RB_XH * RB_XW * a.ps * 16;
    const size_t lds = (size_t)a.tab_bytes + 2 * (size_t)a.w_bytes + 2 * (size_t)a.x_bytes +
                       (size_t)RB_IH * RB_IW * a.ps * 16;
    a.tiles_x = rv_cdiv(w, RB_TW);
    a.n_tiles = a.tiles_x * rv_cdiv(h, RB_TH);
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        RV_HIP(hipGetDevice(&dev));
        RV_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    dim3 grid(a.n_tiles < n_cu ? a.n_tiles : n_cu);             // persistent: one workgroup per CU
    if (MT == 1) return launch_resblock<1>(a, grid, lds, (hipStream_t)stream);
    return launch_resblock<2>(a, grid, lds, (hipStream_t)stream);
}
