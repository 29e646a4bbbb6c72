// Reference matching on CDNA4 matrix cores  (FeatureMatching.forward, RefVSR_/attention.py:72-91).
//
//   corr[r][p] = < ref_patch r , lr_patch p >   (144-dim, both L2-normalised)
//   conf[p], idx[p] = max / argmax over r        (first maximal index wins, like torch.max)
//
// The reference materialises corr ([32400 x 129600] fp32 = 16.8 GB at 270p) with one GEMM and then
// reduces it.  Here the GEMM and the column reduction are fused: a workgroup owns 512 LR columns,
// keeps their fp16 operand fragments in registers for its whole lifetime, streams the reference
// rows through LDS in 128-row chunks (register-prefetched, double buffered) and keeps a running
// top-2 per column in registers -- corr never leaves the accumulators.
//
//  * v_mfma_f32_32x32x16_f16: K = 144 = 9 steps exactly (no K padding).  The 32x32 accumulator
//    layout gives each lane 16 rows of ONE column, so the column reduction is lane-local.
//  * LDS rows are 304 bytes (K padded to 152 halfs on the host side of the ABI): 19 sixteen-byte
//    slots per row, odd => the 16 lanes of a ds_read_b128 group hit 16 distinct slots.
//  * top-2 (not top-1) is kept so that the fp16 operand rounding cannot change the winner: the
//    candidates are re-ranked with an exact fp32 dot product by match_refine.
#include <stdlib.h>

#include "common.h"

#define KP REFVSR_MATCH_KP            // halfs per row (152)
#define ROWB (KP * 2)                 // bytes per row (304)
#define COLB REFVSR_MATCH_COLBLOCK    // LR columns per workgroup
#define KSTEPS 9

// ---------------------------------------------------------------------------------------------
// patch rows: reflect-pad 3x3 unfold + L2 normalise -> fp16 [L][KP], plus 1/norm
// ---------------------------------------------------------------------------------------------
__global__ void match_patches_kernel(const float* __restrict__ feat, int h, int w, f16* __restrict__ rows,
                                     float* __restrict__ inv_norm, float* __restrict__ rows32) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= h * w) return;
    const int y = p / w, x = p - y * w;
    int yy[3], xx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { yy[k] = rv_reflect(y + k - 1, h); xx[k] = rv_reflect(x + k - 1, w); }
    const size_t plane = (size_t)h * w;
    float ss = 0.0f;
    for (int c = 0; c < 16; ++c) {
        const float* f = feat + c * plane;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) { const float v = f[(size_t)yy[ky] * w + xx[kx]]; ss = fmaf(v, v, ss); }
    }
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    inv_norm[p] = inv;
    f16* row = rows + (size_t)p * KP;
    // 144 = 18 groups of 8 halfs; element e = c*9 + ky*3 + kx
    for (int g = 0; g < 19; ++g) {
        f16x8 o;
        float raw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = g * 8 + k;
            float v = 0.0f;
            if (e < 144) {
                const int c = e / 9, t = e - c * 9;
                const int ky = t / 3, kx = t - ky * 3;
                v = feat[c * plane + (size_t)yy[ky] * w + xx[kx]];
            }
            raw[k] = v;
            o[k] = (f16)(v * inv);
        }
        *reinterpret_cast<f16x8*>(row + g * 8) = o;
        if (rows32 && g < 18) {                                // un-normalised fp32 patch (the exact search's operand)
            float4* d = reinterpret_cast<float4*>(rows32 + (size_t)p * 144 + g * 8);
            d[0] = make_float4(raw[0], raw[1], raw[2], raw[3]);
            d[1] = make_float4(raw[4], raw[5], raw[6], raw[7]);
        }
    }
}

extern "C" int refvsr_match_patches(const float* feat, int h, int w, void* rows, float* inv_norm, float* rows32,
                                    void* stream) {
    RV_CHECK(feat && rows && inv_norm && h >= 2 && w >= 2, "match_patches: bad args");
    hipLaunchKernelGGL(match_patches_kernel, dim3(rv_cdiv(h * w, 128)), dim3(128), 0, (hipStream_t)stream,
                       feat, h, w, (f16*)rows, inv_norm, rows32);
    RV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// fused GEMM + column top-2
// ---------------------------------------------------------------------------------------------
struct Top2 { float m1, m2; int i1, i2; };

__device__ __forceinline__ bool rv_better(float va, int ia, float vb, int ib) {
    return va > vb || (va == vb && ia < ib);
}

__device__ __forceinline__ float acc_max(const f32x16& a) {
    float t0 = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), t1 = fmaxf(fmaxf(a[4], a[5]), fmaxf(a[6], a[7]));
    float t2 = fmaxf(fmaxf(a[8], a[9]), fmaxf(a[10], a[11])), t3 = fmaxf(fmaxf(a[12], a[13]), fmaxf(a[14], a[15]));
    return fmaxf(fmaxf(t0, t1), fmaxf(t2, t3));
}

__device__ __forceinline__ void top2_insert(Top2& s, const f32x16& acc, int rowbase, int n_ref) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {                      // increasing r == increasing row index; branch-free selects
        const int row = rowbase + (r & 3) + 8 * (r >> 2);
        const float v = (row < n_ref) ? acc[r] : -INFINITY;
        const bool g1 = v > s.m1;
        const bool g2 = v > s.m2;
        s.m2 = g1 ? s.m1 : (g2 ? v : s.m2);
        s.i2 = g1 ? s.i1 : (g2 ? row : s.i2);
        s.m1 = g1 ? v : s.m1;
        s.i1 = g1 ? row : s.i1;
    }
}

// Schedule history (round 1, identical outputs, MI355X): plain loop 2.0 ms -> prefetch pinned around the MFMA loop 1.95 ->
// A-fragment double buffering 1.69 -> accumulator double buffering + branch-free top-2 (the VALU max-trees run under the
// matrix pipe) 1.26 -> 256-row stages (below) 1.17 ms.  The earlier variants were removed from the library in round 2.
// Accumulator double buffering on 256-row stages: one barrier per 288
// MFMAs/wave, LDS 2 x 76 KiB, the next stage fetched in two halves so only 20 VGPRs are pinned.
// (Keeping a second A-fragment set in flight as well needs > 256 VGPRs at 2 waves/SIMD and spills.)
#define CHUNK4 REFVSR_MATCH_ROWCHUNK
#define CHUNK4_U4 (CHUNK4 * ROWB / 16)
#define PF4 ((CHUNK4_U4 + 511) / 512)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void match_top2_kernel_v4(
    const f16* __restrict__ ref_rows, int n_ref, const f16* __restrict__ lr_rows, int n_lr,
    int chunks_per_split, int n_chunks, int row_splits, int32_t* __restrict__ cand_idx, float* __restrict__ cand_val) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][CHUNK4 * ROWB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int col0 = blockIdx.x * COLB + wave * 64;
    const int c_begin = blockIdx.y * chunks_per_split;
    const int c_end = min(c_begin + chunks_per_split, n_chunks);

    f16x8 bfrag[2][KSTEPS];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const f16* src = lr_rows + (size_t)(col0 + ct * 32 + l31) * KP + hi * 8;
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) bfrag[ct][k] = *reinterpret_cast<const f16x8*>(src + k * 16);
    }
    Top2 st[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) { st[ct].m1 = st[ct].m2 = -INFINITY; st[ct].i1 = st[ct].i2 = 0; }

    const uint4* gsrc = reinterpret_cast<const uint4*>(ref_rows);
    // the next stage is fetched in two halves (HALF_U4 uint4 each) so only PF4H x 4 VGPRs are pinned
    constexpr int HALF_U4 = CHUNK4_U4 / 2;                       // 2432
    constexpr int PF4H = (HALF_U4 + 511) / 512;                  // 5
    static_assert(PF4H == 5, "prefetch macros assume 5 slots");
    uint4 pf0, pf1, pf2, pf3, pf4;
#define PF_LOAD(base) do { pf0 = gsrc[(base) + pfi[0]]; pf1 = gsrc[(base) + pfi[1]]; pf2 = gsrc[(base) + pfi[2]]; \
                           pf3 = gsrc[(base) + pfi[3]]; pf4 = gsrc[(base) + pfi[4]]; } while (0)
#define PF_STORE(dst, off) do { uint4* d_ = reinterpret_cast<uint4*>(dst) + (off); d_[pfi[0]] = pf0; d_[pfi[1]] = pf1; \
                                d_[pfi[2]] = pf2; d_[pfi[3]] = pf3; if (last_ok) d_[pfi[4]] = pf4; } while (0)
    int pfi[PF4H];
#pragma unroll
    for (int k = 0; k < PF4H; ++k) pfi[k] = min(tid + k * 512, HALF_U4 - 1);
    const bool last_ok = (tid + (PF4H - 1) * 512) < HALF_U4;

    if (c_begin < c_end) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            PF_LOAD((size_t)c_begin * CHUNK4_U4 + hf * HALF_U4);
            PF_STORE(lds[0], hf * HALF_U4);
        }
    }
    __syncthreads();

    constexpr int NRT = CHUNK4 / 32;
    int buf = 0;
    for (int c = c_begin; c < c_end; ++c) {
        const bool has_next = (c + 1 < c_end);
        if (has_next) PF_LOAD((size_t)(c + 1) * CHUNK4_U4);
        asm volatile("" ::: "memory");
        const unsigned char* ap = lds[buf] + (size_t)l31 * ROWB + hi * 16;
        f16x8 afA[KSTEPS], afB[KSTEPS];
        f32x16 accA0, accA1, accB0, accB1;
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) afA[k] = *reinterpret_cast<const f16x8*>(ap + k * 32);
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA0[r] = 0.0f; accA1[r] = 0.0f; }
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) {
            accA0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afA[k], bfrag[0][k], accA0, 0, 0, 0);
            accA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afA[k], bfrag[1][k], accA1, 0, 0, 0);
        }
        auto run_pairs = [&](const int rp_begin, const int rp_end) {
#pragma unroll 1
        for (int rp = rp_begin; rp < rp_end; ++rp) {         // two tiles per iteration: roles of the A/B sets are static
            const bool more = (rp + 1 < NRT / 2);
            const unsigned char* an = ap + (size_t)(2 * rp + 2) * 32 * ROWB;
            // -- tile e = 2rp is finished (or in flight) in accA; start tile o = 2rp+1
#pragma unroll
            for (int k = 0; k < KSTEPS; ++k) afB[k] = *reinterpret_cast<const f16x8*>(an - (size_t)32 * ROWB + k * 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) { accB0[r] = 0.0f; accB1[r] = 0.0f; }
#pragma unroll
            for (int k = 0; k < KSTEPS; ++k) {
                accB0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afB[k], bfrag[0][k], accB0, 0, 0, 0);
                accB1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afB[k], bfrag[1][k], accB1, 0, 0, 0);
            }
            {
                const float t0 = acc_max(accA0), t1 = acc_max(accA1);
                if (t0 > st[0].m2 || t1 > st[1].m2) {
                    const int rowbase = c * CHUNK4 + (2 * rp) * 32 + 4 * hi;
                    if (t0 > st[0].m2) top2_insert(st[0], accA0, rowbase, n_ref);
                    if (t1 > st[1].m2) top2_insert(st[1], accA1, rowbase, n_ref);
                }
            }
            // -- tile o is in flight in accB; start tile e+2
            if (more) {
#pragma unroll
                for (int k = 0; k < KSTEPS; ++k) afA[k] = *reinterpret_cast<const f16x8*>(an + k * 32);
#pragma unroll
                for (int r = 0; r < 16; ++r) { accA0[r] = 0.0f; accA1[r] = 0.0f; }
#pragma unroll
                for (int k = 0; k < KSTEPS; ++k) {
                    accA0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afA[k], bfrag[0][k], accA0, 0, 0, 0);
                    accA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afA[k], bfrag[1][k], accA1, 0, 0, 0);
                }
            }
            {
                const float t0 = acc_max(accB0), t1 = acc_max(accB1);
                if (t0 > st[0].m2 || t1 > st[1].m2) {
                    const int rowbase = c * CHUNK4 + (2 * rp + 1) * 32 + 4 * hi;
                    if (t0 > st[0].m2) top2_insert(st[0], accB0, rowbase, n_ref);
                    if (t1 > st[1].m2) top2_insert(st[1], accB1, rowbase, n_ref);
                }
            }
        }
        };
        run_pairs(0, NRT / 4);
        asm volatile("" ::: "memory");        // mid-stage: park the first half of the next stage, fetch the second
        if (has_next) {
            PF_STORE(lds[buf ^ 1], 0);
            PF_LOAD((size_t)(c + 1) * CHUNK4_U4 + HALF_U4);
        }
        asm volatile("" ::: "memory");
        run_pairs(NRT / 4, NRT / 2);
        asm volatile("" ::: "memory");
        if (has_next) PF_STORE(lds[buf ^ 1], HALF_U4);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        Top2 a = st[ct], b;
        b.m1 = __shfl_xor(a.m1, 32); b.m2 = __shfl_xor(a.m2, 32);
        b.i1 = __shfl_xor(a.i1, 32); b.i2 = __shfl_xor(a.i2, 32);
        Top2 o;
        if (rv_better(a.m1, a.i1, b.m1, b.i1)) {
            o.m1 = a.m1; o.i1 = a.i1;
            if (rv_better(a.m2, a.i2, b.m1, b.i1)) { o.m2 = a.m2; o.i2 = a.i2; } else { o.m2 = b.m1; o.i2 = b.i1; }
        } else {
            o.m1 = b.m1; o.i1 = b.i1;
            if (rv_better(b.m2, b.i2, a.m1, a.i1)) { o.m2 = b.m2; o.i2 = b.i2; } else { o.m2 = a.m1; o.i2 = a.i1; }
        }
        const int col = col0 + ct * 32 + l31;
        if (hi == 0 && col < n_lr) {
            const size_t o2 = ((size_t)col * row_splits + blockIdx.y) * 2;
            cand_idx[o2] = o.i1; cand_idx[o2 + 1] = o.i2;
            cand_val[o2] = o.m1; cand_val[o2 + 1] = o.m2;
        }
    }
}

#undef PF_LOAD
#undef PF_STORE

extern "C" int refvsr_match_top2(const void* ref_rows, int n_ref, const void* lr_rows, int n_lr, int row_splits,
                                 int32_t* cand_idx, float* cand_val, void* stream) {
    RV_CHECK(ref_rows && lr_rows && cand_idx && cand_val && n_ref >= 2 && n_lr >= 1 && row_splits >= 1,
             "match_top2: bad args");
    const int n_chunks = rv_cdiv(n_ref, CHUNK4);
    RV_CHECK(row_splits <= n_chunks, "match_top2: row_splits (%d) > row chunks (%d)", row_splits, n_chunks);
    const int cps = rv_cdiv(n_chunks, row_splits);
    RV_CHECK((row_splits - 1) * cps < n_chunks, "match_top2: empty row split (use fewer splits)");
    dim3 grid(rv_cdiv(n_lr, COLB), row_splits);
    hipLaunchKernelGGL(match_top2_kernel_v4, grid, dim3(512), 0, (hipStream_t)stream, (const f16*)ref_rows, n_ref,
                       (const f16*)lr_rows, n_lr, cps, n_chunks, row_splits, cand_idx, cand_val);
    RV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// exact fp32 re-rank of the candidates
// ---------------------------------------------------------------------------------------------
// Summation order of the exact correlation: position p = 16 S + 4 j + q  <->  patch element e = 16 S + 4 q + j (S = 0..8,
// j, q = 0..3).  It is the order in which v_mfma_f32_16x16x4_f32 consumes K when lane group q holds four CONSECUTIVE
// elements of a row (one 16-byte load) and MFMA step (S, j) takes component j -- so the exhaustive search
// (match_exact_kernel) and this re-rank compute bit-identical values.
__device__ __forceinline__ float patch_dot(const float* __restrict__ lf, int h, int w, const int* ly, const int* lx,
                                           const float* __restrict__ rf, int hr, int wr, int ry, int rx) {
    int yy[3], xx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { yy[k] = rv_reflect(ry + k - 1, hr); xx[k] = rv_reflect(rx + k - 1, wr); }
    const size_t lp = (size_t)h * w, rp = (size_t)hr * wr;
    float d = 0.0f;
#pragma unroll 1
    for (int S = 0; S < 9; ++S)                      // (a rolled loop: fully unrolled, the 144 addresses spill)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = 16 * S + 4 * q + j;
                const int c = e / 9, t = e - c * 9;
                const int ky = t / 3, kx = t - ky * 3;
                d = fmaf(lf[c * lp + (size_t)ly[ky] * w + lx[kx]], rf[c * rp + (size_t)yy[ky] * wr + xx[kx]], d);
            }
    return d;
}

__global__ void match_refine_kernel(const float* __restrict__ lf, int h, int w, const float* __restrict__ rf, int hr,
                                    int wr, const float* __restrict__ inv_lr, const float* __restrict__ inv_ref,
                                    const int32_t* __restrict__ cand, const float* __restrict__ cand_val, int ncand,
                                    float margin, int32_t* __restrict__ flagged, float* __restrict__ conf,
                                    int32_t* __restrict__ idx) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= h * w) return;
    const int y = p / w, x = p - y * w;
    int ly[3], lx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { ly[k] = rv_reflect(y + k - 1, h); lx[k] = rv_reflect(x + k - 1, w); }
    const float il = inv_lr[p];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    const int n_ref = hr * wr;
    for (int k = 0; k < ncand; ++k) {
        int r = cand[(size_t)p * ncand + k];
        r = min(max(r, 0), n_ref - 1);
        const int ry = r / wr, rx = r - ry * wr;
        const float v = patch_dot(lf, h, w, ly, lx, rf, hr, wr, ry, rx) * il * inv_ref[r];
        if (v > best || (v == best && r < bi)) { best = v; bi = r; }
    }
    conf[p] = best;
    idx[p] = bi;
    // Can a row OUTSIDE the candidate list beat `best`?  Its fp16-GEMM score is <= the runner-up's (the second entry of
    // every row split), its exact value at most `margin` above that: if best clears that bound the answer is final,
    // otherwise the column goes to the exhaustive fp32 search (refvsr_match_exact).
    if (flagged) {
        float m2 = -INFINITY;
        for (int k = 1; k < ncand; k += 2) m2 = fmaxf(m2, cand_val[(size_t)p * ncand + k]);
        if (!(best - m2 >= margin)) flagged[1 + atomicAdd(flagged, 1)] = p;
    }
}

extern "C" int refvsr_match_refine(const float* lr_feat, int h, int w, const float* ref_feat, int hr, int wr,
                                   const float* inv_lr, const float* inv_ref, const int32_t* cand_idx,
                                   const float* cand_val, int ncand, float margin, int32_t* flagged, float* conf,
                                   int32_t* idx, void* stream) {
    RV_CHECK(lr_feat && ref_feat && inv_lr && inv_ref && cand_idx && conf && idx, "match_refine: null pointer");
    RV_CHECK(h >= 2 && w >= 2 && hr >= 2 && wr >= 2 && ncand >= 1, "match_refine: bad sizes");
    RV_CHECK(flagged == nullptr || (cand_val != nullptr && ncand % 2 == 0), "match_refine: flagging needs the top-2 values");
    hipLaunchKernelGGL(match_refine_kernel, dim3(rv_cdiv(h * w, 128)), dim3(128), 0, (hipStream_t)stream,
                       lr_feat, h, w, ref_feat, hr, wr, inv_lr, inv_ref, cand_idx, cand_val, ncand, margin, flagged, conf, idx);
    RV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// exhaustive exact-fp32 search for the columns the fp16 GEMM cannot decide
// ---------------------------------------------------------------------------------------------
// The fp16 operands of match_top2 perturb a correlation by ~3e-5 (up to ~1e-4 for peaky patches).  Where the best two
// candidates are further apart than that, the top-2 + fp32 re-rank is provably the exact arg-max; where they are not
// (flat or repetitive image regions: many reference patches nearly equally similar) a THIRD row may be the true maximum.
// Those columns -- flagged by match_refine -- are searched exhaustively here on v_mfma_f32_16x16x4_f32 (bitwise an fp32
// FMA chain over the 144 patch elements in the order documented at patch_dot, i.e. the same value patch_dot computes): 64 flagged columns per workgroup pass,
// the reference rows (fp32 [n_ref][144], written by match_patches) split over the 4 waves x gridDim.y; per column
// (value, first index) maxima are merged through a 64-bit atomicMax key.  The list is read on the device: no host sync.
#define EX_COLS 64
__device__ __forceinline__ unsigned long long ex_key(float v, int row) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                // order-preserving float -> uint
    return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)row);   // ties: smaller row wins
}

__global__ __launch_bounds__(256) void match_exact_kernel(const float* __restrict__ lf, int h, int w,
                                                          const float* __restrict__ ref32, int n_ref,
                                                          const float* __restrict__ inv_lr, const float* __restrict__ inv_ref,
                                                          const int32_t* __restrict__ flagged,
                                                          unsigned long long* __restrict__ keys) {
    const int count = flagged[0];
    const int ngroups = (count + EX_COLS - 1) / EX_COLS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    const int n_tiles = (n_ref + 15) >> 4;
    // work item = (column group, row part): the reference rows are split so that there are about as many items as
    // workgroups (the flagged count is only known here, on the device)
    const int nsplit = max(1, min(64, (int)gridDim.x / max(ngroups, 1)));
    const int nparts = nsplit * 4;
    const size_t plane = (size_t)h * w;
    for (int item = blockIdx.x; item < ngroups * nsplit; item += gridDim.x) {
        const int g = item / nsplit;
        const int part = (item - g * nsplit) * 4 + wave;
        // B operand: 4 tiles of 16 flagged columns; MFMA step (S, j) takes patch element 16 S + 4 kq + j from lane group kq
        float b[4][36];
        float il[4];
        int colv[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int fi = g * EX_COLS + ct * 16 + n16;
            const int col = fi < count ? flagged[1 + fi] : -1;
            colv[ct] = col;
            const int cc = max(col, 0);
            const int y = cc / w, x = cc - y * w;
            il[ct] = col >= 0 ? inv_lr[cc] : 0.0f;
#pragma unroll
            for (int s = 0; s < 36; ++s) {
                const int e = 16 * (s >> 2) + 4 * kq + (s & 3);
                const int c = e / 9, t = e - c * 9;
                const int ky = t / 3, kx = t - ky * 3;
                b[ct][s] = lf[c * plane + (size_t)rv_reflect(y + ky - 1, h) * w + rv_reflect(x + kx - 1, w)];
            }
        }
        float best[4];
        int besti[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { best[ct] = -INFINITY; besti[ct] = 0x7fffffff; }
        for (int tile = part; tile < n_tiles; tile += nparts) {
            // A operand: lane (row n16, group kq) reads 16 bytes = elements 16 S + 4 kq .. + 3 of its row, nine times
            const float4* arow = reinterpret_cast<const float4*>(ref32 + (size_t)min(tile * 16 + n16, n_ref - 1) * 144 + 4 * kq);
            float4 a[9];
#pragma unroll
            for (int S = 0; S < 9; ++S) a[S] = arow[S * 4];
            f32x4 acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < 9; ++S) {
                const float av[4] = {a[S].x, a[S].y, a[S].z, a[S].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b[ct][S * 4 + j], acc[ct], 0, 0, 0);
            }
            // lane holds rows 4*kq + j (j = 0..3) of column n16: increasing j == increasing row, strict > keeps the first
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = tile * 16 + 4 * kq + j;
                if (row < n_ref) {
                    const float ir = inv_ref[row];
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const float v = acc[ct][j] * il[ct] * ir;
                        if (v > best[ct]) { best[ct] = v; besti[ct] = row; }
                    }
                }
            }
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            unsigned long long k = besti[ct] == 0x7fffffff ? 0ull : ex_key(best[ct], besti[ct]);
            const unsigned long long k1 = __shfl_xor(k, 16);
            k = k1 > k ? k1 : k;
            const unsigned long long k2 = __shfl_xor(k, 32);
            k = k2 > k ? k2 : k;
            if (kq == 0 && colv[ct] >= 0 && k != 0ull) atomicMax(&keys[g * EX_COLS + ct * 16 + n16], k);
        }
    }
}

__global__ void match_exact_finish_kernel(const int32_t* __restrict__ flagged, const unsigned long long* __restrict__ keys,
                                          float* __restrict__ conf, int32_t* __restrict__ idx) {
    const int count = flagged[0];
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < count; f += gridDim.x * blockDim.x) {
        const unsigned long long k = keys[f];
        unsigned u = (unsigned)(k >> 32);
        u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
        const int col = flagged[1 + f];
        conf[col] = __uint_as_float(u);
        idx[col] = (int)(0xffffffffu - (unsigned)(k & 0xffffffffull));
    }
}

extern "C" int refvsr_match_exact(const float* lr_feat, int h, int w, const float* ref_rows32, int n_ref,
                                  const float* inv_lr, const float* inv_ref, const int32_t* flagged, void* keys,
                                  float* conf, int32_t* idx, void* stream) {
    RV_CHECK(lr_feat && ref_rows32 && inv_lr && inv_ref && flagged && keys && conf && idx, "match_exact: null pointer");
    RV_CHECK(h >= 2 && w >= 2 && n_ref >= 2, "match_exact: bad sizes");
    // fixed grid, one workgroup per CU (the flagged count lives on the device; the kernel sizes its work items from it)
    hipLaunchKernelGGL(match_exact_kernel, dim3(rv_num_cus()), dim3(256), 0, (hipStream_t)stream, lr_feat, h, w, ref_rows32, n_ref,
                       inv_lr, inv_ref, flagged, (unsigned long long*)keys);
    RV_LAUNCH_CHECK();
    hipLaunchKernelGGL(match_exact_finish_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, flagged,
                       (const unsigned long long*)keys, conf, idx);
    RV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// unfused fp32 exhaustive search (debugging / cross-check only)
// ---------------------------------------------------------------------------------------------
__global__ void match_naive_kernel(const float* __restrict__ lf, int h, int w, const float* __restrict__ rf, int hr,
                                   int wr, float* __restrict__ conf, int32_t* __restrict__ idx) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= h * w) return;
    const int y = p / w, x = p - y * w;
    int ly[3], lx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { ly[k] = rv_reflect(y + k - 1, h); lx[k] = rv_reflect(x + k - 1, w); }
    const float nl = patch_dot(lf, h, w, ly, lx, lf, h, w, y, x);
    const float il = 1.0f / fmaxf(sqrtf(nl), 1e-12f);
    float best = -INFINITY;
    int bi = 0;
    for (int r = 0; r < hr * wr; ++r) {
        const int ry = r / wr, rx = r - ry * wr;
        int yy[3], xx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { yy[k] = rv_reflect(ry + k - 1, hr); xx[k] = rv_reflect(rx + k - 1, wr); }
        const float nr = patch_dot(rf, hr, wr, yy, xx, rf, hr, wr, ry, rx);
        const float v = patch_dot(lf, h, w, ly, lx, rf, hr, wr, ry, rx) * il * (1.0f / fmaxf(sqrtf(nr), 1e-12f));
        if (v > best) { best = v; bi = r; }
    }
    conf[p] = best;
    idx[p] = bi;
}

extern "C" int refvsr_match_naive(const float* lr_feat, int h, int w, const float* ref_feat, int hr, int wr,
                                  float* conf, int32_t* idx, void* stream) {
    RV_CHECK(lr_feat && ref_feat && conf && idx && h >= 2 && w >= 2 && hr >= 2 && wr >= 2, "match_naive: bad args");
    RV_CHECK((long long)h * w * hr * wr <= (1ll << 31), "match_naive: problem too large for the debug kernel");
    hipLaunchKernelGGL(match_naive_kernel, dim3(rv_cdiv(h * w, 64)), dim3(64), 0, (hipStream_t)stream,
                       lr_feat, h, w, ref_feat, hr, wr, conf, idx);
    RV_LAUNCH_CHECK();
    return 0;
}
