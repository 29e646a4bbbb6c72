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
    unsigned long long* probe;           // debug: per-workgroup s_memtime stamps (refvsr_set_probe), normally null
    int probe_iter;                      // which tile iteration of the workgroup is stamped
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
#define RB_STAMP(i) do { if (p.probe && tid == 0) p.probe[blockIdx.x * 12 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    RB_STAMP(0);
    if (p.probe && tid == 0) p.probe[blockIdx.x * 12 + 8] = __builtin_amdgcn_s_memrealtime();

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
    RB_STAMP(1);
    __syncthreads();
    RB_STAMP(2);

    int cur = 0, iter = 0;
    for (; tile < p.n_tiles; tile += gridDim.x, ++iter) {
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
            if (iter == p.probe_iter) RB_STAMP(3);
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
        if (iter == p.probe_iter) RB_STAMP(4);
        __syncthreads();
        if (iter == p.probe_iter) RB_STAMP(5);

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
            if (iter == p.probe_iter) RB_STAMP(6);
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
        if (iter == p.probe_iter) RB_STAMP(7);
        __syncthreads();                   // t tile free for the next phase 1
        cur ^= 1;
    }
    if (p.probe && tid == 0) {
        p.probe[blockIdx.x * 12 + 9] = __builtin_amdgcn_s_memrealtime();
        p.probe[blockIdx.x * 12 + 10] = __builtin_amdgcn_s_memtime();
    }
#undef RB_STAMP
}

unsigned long long* g_rb_probe = nullptr;          // shared with resblock_lean.hip
int g_rb_probe_iter = 0;
extern "C" int refvsr_set_probe(void* buf, int iter) {
    g_rb_probe = (unsigned long long*)buf;
    g_rb_probe_iter = iter;
    return 0;
}

template <int MT>
static int launch_resblock(const ResBlockArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done[RV_MAX_DEVICES] = {};
    const int dev = rv_device();
    if (!attr_done[dev]) {
        RV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_mfma_kernel<MT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((resblock_mfma_kernel<MT>), grid, dim3(512), lds, st, a);
    RV_LAUNCH_CHECK();
    return 0;
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
    a.probe = g_rb_probe;
    a.probe_iter = g_rb_probe_iter;
    const int n_cu = rv_num_cus();
    dim3 grid(a.n_tiles < n_cu ? a.n_tiles : n_cu);             // persistent: one workgroup per CU
    if (MT == 1) return launch_resblock<1>(a, grid, lds, (hipStream_t)stream);
    return launch_resblock<2>(a, grid, lds, (hipStream_t)stream);
}
