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
#define CHUNK 128                     // reference rows per LDS stage (variants 0-3); variant 4 stages REFVSR_MATCH_ROWCHUNK
#define COLB REFVSR_MATCH_COLBLOCK    // LR columns per workgroup
#define KSTEPS 9
#define CHUNK_U4 (CHUNK * ROWB / 16)  // 2432 uint4 per stage
#define PF ((CHUNK_U4 + 511) / 512)   // uint4 prefetch registers per thread (5)

// ---------------------------------------------------------------------------------------------
// patch rows: reflect-pad 3x3 unfold + L2 normalise -> fp16 [L][KP], plus 1/norm
// ---------------------------------------------------------------------------------------------
__global__ void match_patches_kernel(const float* __restrict__ feat, int h, int w, f16* __restrict__ rows,
                                     float* __restrict__ inv_norm) {
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
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = g * 8 + k;
            float v = 0.0f;
            if (e < 144) {
                const int c = e / 9, t = e - c * 9;
                const int ky = t / 3, kx = t - ky * 3;
                v = feat[c * plane + (size_t)yy[ky] * w + xx[kx]] * inv;
            }
            o[k] = (f16)v;
        }
        *reinterpret_cast<f16x8*>(row + g * 8) = o;
    }
}

extern "C" int refvsr_match_patches(const float* feat, int h, int w, void* rows, float* inv_norm, void* stream) {
    RV_CHECK(feat && rows && inv_norm && h >= 2 && w >= 2, "match_patches: bad args");
    hipLaunchKernelGGL(match_patches_kernel, dim3(rv_cdiv(h * w, 128)), dim3(128), 0, (hipStream_t)stream,
                       feat, h, w, (f16*)rows, inv_norm);
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

__device__ __forceinline__ void top2_scan(Top2& s, const f32x16& acc, int rowbase, int n_ref) {
    // fast reject: nothing in this 16-row slice beats the running second best
    float tm = fmaxf(fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])), fmaxf(fmaxf(acc[4], acc[5]), fmaxf(acc[6], acc[7])));
    tm = fmaxf(tm, fmaxf(fmaxf(fmaxf(acc[8], acc[9]), fmaxf(acc[10], acc[11])), fmaxf(fmaxf(acc[12], acc[13]), fmaxf(acc[14], acc[15]))));
    if (tm > s.m2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {                  // increasing r == increasing row index
            const int row = rowbase + (r & 3) + 8 * (r >> 2);
            const float v = acc[r];
            if (row < n_ref) {
                if (v > s.m1) { s.m2 = s.m1; s.i2 = s.i1; s.m1 = v; s.i1 = row; }
                else if (v > s.m2) { s.m2 = v; s.i2 = row; }
            }
        }
    }
}

// V = 0: first version (kept for A/B: the compiler hoists the LDS write of the prefetched chunk above
//        the compute loop, exposing the global-load latency once per chunk).
// V = 1: prefetch pinned -- global loads issued before, LDS writes after the MFMA loop.
// V = 2: V1 + row-tile loop fully unrolled with the next tile's A fragments read during the current
//        tile's MFMAs (explicit register double buffering).
// V = 3: V2 + accumulator double buffering: the 18 MFMAs of row tile rt+1 are issued BEFORE the column
//        reduction of tile rt, so the VALU max-trees run under the matrix pipe; the two per-column-tile
//        slow paths share one (rare) branch.
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

template <int V>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void match_top2_kernel(
    const f16* __restrict__ ref_rows, int n_ref, const f16* __restrict__ lr_rows, int n_lr,
    int chunks_per_split, int n_chunks, int row_splits, int32_t* __restrict__ cand_idx, float* __restrict__ cand_val) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][CHUNK * ROWB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int col0 = blockIdx.x * COLB + wave * 64;
    const int c_begin = blockIdx.y * chunks_per_split;
    const int c_end = min(c_begin + chunks_per_split, n_chunks);

    // stationary operand: this wave's 2 x 32 LR columns, 9 K-steps each (72 VGPRs)
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
    uint4 pf[PF];
    // per-thread staging slots: slot k covers uint4 index tid + 512*k; the last one is clamped (branch-free load)
    int pfi[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) pfi[k] = min(tid + k * 512, CHUNK_U4 - 1);
    const bool last_ok = (tid + (PF - 1) * 512) < CHUNK_U4;

    if (c_begin < c_end) {                     // prologue: first chunk straight to LDS buffer 0
#pragma unroll
        for (int k = 0; k < PF; ++k) pf[k] = gsrc[(size_t)c_begin * CHUNK_U4 + pfi[k]];
#pragma unroll
        for (int k = 0; k < PF; ++k)
            if (k < PF - 1 || last_ok) reinterpret_cast<uint4*>(lds[0])[pfi[k]] = pf[k];
    }
    __syncthreads();

    int buf = 0;
    for (int c = c_begin; c < c_end; ++c) {
        const bool has_next = (c + 1 < c_end);
        if (has_next) {
#pragma unroll
            for (int k = 0; k < PF; ++k) pf[k] = gsrc[(size_t)(c + 1) * CHUNK_U4 + pfi[k]];
        }
        if (V >= 1) asm volatile("" ::: "memory");        // keep the loads above, in flight during the MFMAs
        const unsigned char* L = lds[buf];
        if (V == 3) {
            constexpr int NRT = CHUNK / 32;
            f16x8 afA[KSTEPS], afB[KSTEPS];
            const unsigned char* ap = L + (size_t)l31 * ROWB + hi * 16;
#pragma unroll
            for (int k = 0; k < KSTEPS; ++k) afA[k] = *reinterpret_cast<const f16x8*>(ap + k * 32);
            f32x16 accA0, accA1, accB0, accB1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { accA0[r] = 0.0f; accA1[r] = 0.0f; }
#pragma unroll
            for (int k = 0; k < KSTEPS; ++k) {
                accA0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afA[k], bfrag[0][k], accA0, 0, 0, 0);
                accA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afA[k], bfrag[1][k], accA1, 0, 0, 0);
            }
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                f32x16& c0 = (rt & 1) ? accB0 : accA0;           // finished tile
                f32x16& c1 = (rt & 1) ? accB1 : accA1;
                f32x16& n0 = (rt & 1) ? accA0 : accB0;           // tile in flight
                f32x16& n1 = (rt & 1) ? accA1 : accB1;
                f16x8* nf = (rt & 1) ? afA : afB;
                if (rt + 1 < NRT) {
                    const unsigned char* an = ap + (size_t)(rt + 1) * 32 * ROWB;
#pragma unroll
                    for (int k = 0; k < KSTEPS; ++k) nf[k] = *reinterpret_cast<const f16x8*>(an + k * 32);
#pragma unroll
                    for (int r = 0; r < 16; ++r) { n0[r] = 0.0f; n1[r] = 0.0f; }
#pragma unroll
                    for (int k = 0; k < KSTEPS; ++k) {
                        n0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(nf[k], bfrag[0][k], n0, 0, 0, 0);
                        n1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(nf[k], bfrag[1][k], n1, 0, 0, 0);
                    }
                }
                const float t0 = acc_max(c0), t1 = acc_max(c1);
                if (t0 > st[0].m2 || t1 > st[1].m2) {
                    const int rowbase = c * CHUNK + rt * 32 + 4 * hi;
                    if (t0 > st[0].m2) top2_insert(st[0], c0, rowbase, n_ref);
                    if (t1 > st[1].m2) top2_insert(st[1], c1, rowbase, n_ref);
                }
            }
        } else if (V <= 1) {
#pragma unroll 1
            for (int rt = 0; rt < CHUNK / 32; ++rt) {
                f16x8 afrag[KSTEPS];
                const unsigned char* ap = L + (size_t)(rt * 32 + l31) * ROWB + hi * 16;
#pragma unroll
                for (int k = 0; k < KSTEPS; ++k) afrag[k] = *reinterpret_cast<const f16x8*>(ap + k * 32);
                f32x16 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
#pragma unroll
                for (int k = 0; k < KSTEPS; ++k) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[k], bfrag[0][k], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[k], bfrag[1][k], acc1, 0, 0, 0);
                }
                const int rowbase = c * CHUNK + rt * 32 + 4 * hi;
                top2_scan(st[0], acc0, rowbase, n_ref);
                top2_scan(st[1], acc1, rowbase, n_ref);
            }
        } else {
            f16x8 afA[KSTEPS], afB[KSTEPS];
            const unsigned char* ap = L + (size_t)l31 * ROWB + hi * 16;
#pragma unroll
            for (int k = 0; k < KSTEPS; ++k) afA[k] = *reinterpret_cast<const f16x8*>(ap + k * 32);
#pragma unroll
            for (int rt = 0; rt < CHUNK / 32; ++rt) {
                f16x8* cur = (rt & 1) ? afB : afA;
                f16x8* nxt = (rt & 1) ? afA : afB;
                if (rt + 1 < CHUNK / 32) {
                    const unsigned char* an = ap + (size_t)(rt + 1) * 32 * ROWB;
#pragma unroll
                    for (int k = 0; k < KSTEPS; ++k) nxt[k] = *reinterpret_cast<const f16x8*>(an + k * 32);
                }
                f32x16 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
#pragma unroll
                for (int k = 0; k < KSTEPS; ++k) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[k], bfrag[0][k], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[k], bfrag[1][k], acc1, 0, 0, 0);
                }
                const int rowbase = c * CHUNK + rt * 32 + 4 * hi;
                top2_scan(st[0], acc0, rowbase, n_ref);
                top2_scan(st[1], acc1, rowbase, n_ref);
            }
        }
        if (V >= 1) asm volatile("" ::: "memory");        // ... and the LDS writes below
        if (has_next) {
#pragma unroll
            for (int k = 0; k < PF; ++k)
                if (k < PF - 1 || last_ok) reinterpret_cast<uint4*>(lds[buf ^ 1])[pfi[k]] = pf[k];
        }
        __syncthreads();
        buf ^= 1;
    }

    // merge the two half-waves (same column, disjoint row subsets), then lanes 0..31 publish
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

// Variant 4: variant 3's schedule (accumulator double buffering) on 256-row stages: one barrier per 288
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
    static int variant = -1;                 // tuning knob (A/B of schedules inside one process): REFVSR_MATCH_VARIANT
    const char* ev = getenv("REFVSR_MATCH_VARIANT");
    const int want = ev ? atoi(ev) : 4;
    if (want != variant) variant = (want >= 0 && want <= 4) ? want : 4;
    const int chunk_rows = (variant == 4) ? CHUNK4 : CHUNK;
    const int n_chunks = rv_cdiv(n_ref, chunk_rows);
    RV_CHECK(row_splits <= n_chunks, "match_top2: row_splits (%d) > row chunks (%d)", row_splits, n_chunks);
    const int cps = rv_cdiv(n_chunks, row_splits);
    RV_CHECK((row_splits - 1) * cps < n_chunks, "match_top2: empty row split (use fewer splits)");
    dim3 grid(rv_cdiv(n_lr, COLB), row_splits);
#define RV_MATCH_LAUNCH(V)                                                                                   \
    hipLaunchKernelGGL(match_top2_kernel<V>, grid, dim3(512), 0, (hipStream_t)stream, (const f16*)ref_rows, \
                       n_ref, (const f16*)lr_rows, n_lr, cps, n_chunks, row_splits, cand_idx, cand_val)
    if (variant == 0) RV_MATCH_LAUNCH(0);
    else if (variant == 1) RV_MATCH_LAUNCH(1);
    else if (variant == 2) RV_MATCH_LAUNCH(2);
    else if (variant == 3) RV_MATCH_LAUNCH(3);
    else
        hipLaunchKernelGGL(match_top2_kernel_v4, grid, dim3(512), 0, (hipStream_t)stream, (const f16*)ref_rows, n_ref,
                           (const f16*)lr_rows, n_lr, cps, n_chunks, row_splits, cand_idx, cand_val);
#undef RV_MATCH_LAUNCH
    RV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// exact fp32 re-rank of the candidates
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float patch_dot(const float* __restrict__ lf, int h, int w, const int* ly, const int* lx,
                                           const float* __restrict__ rf, int hr, int wr, int ry, int rx) {
    int yy[3], xx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { yy[k] = rv_reflect(ry + k - 1, hr); xx[k] = rv_reflect(rx + k - 1, wr); }
    const size_t lp = (size_t)h * w, rp = (size_t)hr * wr;
    float d = 0.0f;
    for (int c = 0; c < 16; ++c) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                d = fmaf(lf[c * lp + (size_t)ly[ky] * w + lx[kx]], rf[c * rp + (size_t)yy[ky] * wr + xx[kx]], d);
    }
    return d;
}

__global__ void match_refine_kernel(const float* __restrict__ lf, int h, int w, const float* __restrict__ rf, int hr,
                                    int wr, const float* __restrict__ inv_lr, const float* __restrict__ inv_ref,
                                    const int32_t* __restrict__ cand, int ncand, float* __restrict__ conf,
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
}

extern "C" int refvsr_match_refine(const float* lr_feat, int h, int w, const float* ref_feat, int hr, int wr,
                                   const float* inv_lr, const float* inv_ref, const int32_t* cand_idx, int ncand,
                                   float* conf, int32_t* idx, void* stream) {
    RV_CHECK(lr_feat && ref_feat && inv_lr && inv_ref && cand_idx && conf && idx, "match_refine: null pointer");
    RV_CHECK(h >= 2 && w >= 2 && hr >= 2 && wr >= 2 && ncand >= 1, "match_refine: bad sizes");
    hipLaunchKernelGGL(match_refine_kernel, dim3(rv_cdiv(h * w, 128)), dim3(128), 0, (hipStream_t)stream,
                       lr_feat, h, w, ref_feat, hr, wr, inv_lr, inv_ref, cand_idx, ncand, conf, idx);
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
