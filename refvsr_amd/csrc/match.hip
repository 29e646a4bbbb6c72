// Reference matching on CDNA4 matrix cores  (FeatureMatching.forward, RefVSR_/attention.py:72-91): patch rows, exact
// re-rank and the exhaustive search of ambiguous columns.  The fused GEMM + top-2 kernel lives in match_top2.hip.
//
//   corr[r][p] = < ref_patch r , lr_patch p >   (144-dim, both L2-normalised)
//   conf[p], idx[p] = max / argmax over r        (first maximal index wins, like torch.max)
//
// The reference materialises corr ([32400 x 129600] fp32 = 16.8 GB at 270p) with one GEMM and then
// reduces it.  Here the GEMM and the column reduction are fused: a workgroup owns 512 LR columns,
// keeps their fp16 operand fragments in registers for its whole lifetime, streams the reference
// rows through LDS in 256-row stages (register-prefetched, double buffered) and keeps a running
// top-2 per column in registers -- corr never leaves the accumulators.
//
//  * v_mfma_f32_32x32x16_f16: K = 144 = 9 steps exactly (no K padding).  The 32x32 accumulator
//    layout gives each lane 16 rows of ONE column, so the column reduction is lane-local.
//  * LDS rows are 304 bytes (K padded to 152 halfs on the host side of the ABI): 19 sixteen-byte
//    slots per row, odd => the 16 lanes of a ds_read_b128 group hit 16 distinct slots.
//  * top-2 (not top-1) is kept so that the fp16 operand rounding cannot change the winner: the
//    candidates are re-ranked with an exact fp32 dot product by match_refine, and the columns whose
//    margin is inside the fp16 error are searched exhaustively at fp32 accuracy (match_exact).
#include <stdlib.h>

#include "match_common.h"

// ---------------------------------------------------------------------------------------------
// patch rows: reflect-pad 3x3 unfold + L2 normalise -> fp16 [L][KP], plus 1/norm
// ---------------------------------------------------------------------------------------------
// Scale of the low half of the fp16 hi + lo operand split (exact search): lo = fp16((v - hi) * 2^11), so that it is a
// normal fp16 number wherever hi is (no precision lost to fp16 subnormals).
#define LO_SCALE 2048.0f
#define LO_UNSCALE (1.0f / 2048.0f)

__global__ __launch_bounds__(128) void match_patches_kernel(const float* __restrict__ feat, int h, int w,
                                                            f16* __restrict__ rows, float* __restrict__ inv_norm,
                                                            f16* __restrict__ rows_lo) {
    __shared__ float s_inv[128];
    const int tid = threadIdx.x, p0 = blockIdx.x * 128, n = h * w;
    const size_t plane = (size_t)h * w;
    {   // phase 1: one pixel per thread, 1 / |patch|
        const int p = min(p0 + tid, n - 1);
        const int y = p / w, x = p - y * w;
        int yy[3], xx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { yy[k] = rv_reflect(y + k - 1, h) * w; xx[k] = rv_reflect(x + k - 1, w); }
        float ss = 0.0f;
        for (int c = 0; c < 16; ++c) {
            const float* f = feat + c * plane;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) { const float v = f[yy[ky] + xx[kx]]; ss = fmaf(v, v, ss); }
        }
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        s_inv[tid] = inv;
        if (p0 + tid < n) inv_norm[p0 + tid] = inv;
    }
    __syncthreads();
    // phase 2: the block's 128 rows = 128 x 19 sixteen-byte slots, consecutive threads -> consecutive slots (coalesced
    // stores); slot g of a row holds elements 8 g .. 8 g + 7, element e = c*9 + ky*3 + kx (slot 18 is the zero pad)
    const int n_slots = min(128, n - p0) * 19;
    for (int i = tid; i < n_slots; i += 128) {
        const int pl = i / 19, g = i - pl * 19;
        const int p = p0 + pl;
        const int y = p / w, x = p - y * w;
        const float inv = s_inv[pl];
        f16x8 o, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = g * 8 + k;
            float v = 0.0f;
            if (e < 144) {
                const int c = e / 9, t = e - c * 9;
                const int ky = t / 3, kx = t - ky * 3;
                v = feat[c * plane + (size_t)rv_reflect(y + ky - 1, h) * w + rv_reflect(x + kx - 1, w)];
            }
            float vn = v * inv;
            // (opaque: otherwise hipcc re-derives the hi half for the subtraction with a fused multiply-convert whose
            //  single rounding can differ from the stored double-rounded one by an fp16 ulp)
            asm volatile("" : "+v"(vn));
            o[k] = (f16)vn;
            lo[k] = (f16)((vn - (float)o[k]) * LO_SCALE);
        }
        const size_t off = (size_t)p0 * KP + (size_t)i * 8;
        *reinterpret_cast<f16x8*>(rows + off) = o;
        if (rows_lo) *reinterpret_cast<f16x8*>(rows_lo + off) = lo;    // second half of the split operand
    }
}

extern "C" int refvsr_match_patches(const float* feat, int h, int w, void* rows, float* inv_norm, void* rows_lo,
                                    void* stream) {
    RV_CHECK(feat && rows && inv_norm && h >= 2 && w >= 2, "match_patches: bad args");
    hipLaunchKernelGGL(match_patches_kernel, dim3(rv_cdiv(h * w, 128)), dim3(128), 0, (hipStream_t)stream,
                       feat, h, w, (f16*)rows, inv_norm, (f16*)rows_lo);
    RV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// exact fp32 re-rank of the candidates
// ---------------------------------------------------------------------------------------------
// UNROLL: channels per trip of the (otherwise rolled) channel loop.  Fully unrolled, the 144 addresses spill; rolled, a lone
// thread pays one memory round trip per channel -- match_exact_finish (a few thousand threads, latency-bound) takes 4.
template <int UNROLL = 1>
__device__ __forceinline__ float patch_dot(const float* __restrict__ lf, int h, int w, const int* ly, const int* lx,
                                           const float* __restrict__ rf, int hr, int wr, int ry, int rx) {
    int yy[3], xx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { yy[k] = rv_reflect(ry + k - 1, hr); xx[k] = rv_reflect(rx + k - 1, wr); }
    const size_t lp = (size_t)h * w, rp = (size_t)hr * wr;
    float d = 0.0f;
#pragma unroll 1
    for (int c0 = 0; c0 < 16; c0 += UNROLL) {
        float a[UNROLL][9], b[UNROLL][9];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                a[u][t] = lf[(c0 + u) * lp + (size_t)ly[t / 3] * w + lx[t % 3]];
                b[u][t] = rf[(c0 + u) * rp + (size_t)yy[t / 3] * wr + xx[t % 3]];
            }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int t = 0; t < 9; ++t) d = fmaf(a[u][t], b[u][t], d);     // same order for every UNROLL: c, ky, kx
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
    // (round 6) With the fp16 scores at hand a candidate whose score lies 2 x `margin` or more below the best fp16 score cannot be the
    // exact maximum under the error bound the flagging below rests on (|exact - fp16| <= margin per score: its exact value is at most
    // top16 - margin, the top candidate's at least that) -- so its exact value is not evaluated: most columns have a clear winner and
    // gather ONE reference patch instead of two.  Same winner, same value (the winner's own fp32 expression), bit for bit:
    // tests/test_gpu_ops.py::test_match_*, the full-size index-map fixtures.
    float top16 = -INFINITY;
    if (cand_val)
        for (int k = 0; k < ncand; ++k) top16 = fmaxf(top16, cand_val[(size_t)p * ncand + k]);
    for (int k = 0; k < ncand; ++k) {
        if (cand_val && !(cand_val[(size_t)p * ncand + k] > top16 - 2.0f * margin)) continue;
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
// exhaustive fp32-grade search for the columns the fp16 GEMM cannot decide
// ---------------------------------------------------------------------------------------------
// The fp16 operands of match_top2 perturb a correlation by ~3e-5 (up to ~1e-4 for peaky patches).  Where the best two
// candidates are further apart than that, the top-2 + fp32 re-rank is provably the exact arg-max; where they are not
// (flat or repetitive image regions: many reference patches nearly equally similar) a THIRD row may be the true maximum.
// Those columns -- flagged by match_refine -- are searched exhaustively here with both operands split into fp16
// hi + lo (lo scaled by 2^11): <a, b> = <ah, bh> + 2^-11 (<ah, bl> + <al, bh>), three fp16 MFMAs with fp32 accumulation,
// dropped term 2^-22 -- the accuracy of an fp32 dot product at 4.8x the rate of v_mfma_f32_16x16x4_f32.
// Workgroup item = 256 flagged columns (64 per wave, B operand = their LR patch rows, in registers) x one part of the
// reference rows, which stream through LDS in 64-row stages shared by the four waves; K = 144 = 4 x 32 + 16
// (v_mfma_f32_16x16x32_f16 x 4 + v_mfma_f32_16x16x16_f16).  Per column the (value, first index) maxima are merged through
// a 64-bit atomicMax key; match_exact_finish re-evaluates the winner with patch_dot (the SAME fp32 expression
// match_refine uses) and keeps it only if it beats the re-ranked candidates, so flagged and unflagged columns carry
// values of one definition.  The flagged list is read on the device: no host synchronisation.
#define EX_CT 4                     // 16-column tiles per wave
#define EX_COLS (64 * EX_CT)         // flagged columns per workgroup item (4 waves)
#define EX_ROWS 64
__device__ __forceinline__ unsigned long long ex_key(float v, int row) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                // order-preserving float -> uint
    return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)row);   // ties: smaller row wins
}

#define EX_STAGE_BYTES (EX_ROWS * ROWB)           // one array (hi or lo) of one stage: 19 x 1 KiB
#define EX_CHUNKS (2 * EX_STAGE_BYTES / 1024)     // 1 KiB wave-chunks per stage (hi then lo)

// Stage st of the reference rows (64 rows x 304 bytes, hi and lo arrays) -> LDS, asynchronously: global_load_lds writes
// wave-uniform LDS base + lane * 16, i.e. a flat copy of 1 KiB per wave instruction -- exactly this layout (the LDS image
// of a stage IS its global image).  No staging registers; completion is tracked by vmcnt.
__device__ __forceinline__ void ex_issue_stage(const f16* __restrict__ ref_hi, const f16* __restrict__ ref_lo, int st,
                                               unsigned char* buf, int wave, int lane) {
    const unsigned char* gh = reinterpret_cast<const unsigned char*>(ref_hi + (size_t)st * EX_ROWS * KP);
    const unsigned char* gl = reinterpret_cast<const unsigned char*>(ref_lo + (size_t)st * EX_ROWS * KP);
#pragma unroll
    for (int i = 0; i < (EX_CHUNKS + 3) / 4; ++i) {
        const int c = wave + 4 * i;
        if (c < EX_CHUNKS) {
            const bool lo = c >= EX_CHUNKS / 2;
            const int cc = lo ? c - EX_CHUNKS / 2 : c;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)((lo ? gl : gh) + cc * 1024 + lane * 16),
                (__attribute__((address_space(3))) void*)(buf + c * 1024), 16, 0, 0);
        }
    }
}

// 64 staged rows x (EX_CT x 16) columns of one wave: 4 row tiles x EX_CT x 15 MFMAs; running (value, first row) maxima.
__device__ __forceinline__ void ex_compute_stage(const unsigned char* buf, int st, int n_ref, int n16, int kq,
                                                 const f16x8 (&bh)[EX_CT][4], const f16x8 (&bl)[EX_CT][4],
                                                 const f16x4 (&bht)[EX_CT], const f16x4 (&blt)[EX_CT],
                                                 float (&best)[EX_CT], int (&besti)[EX_CT]) {
#pragma unroll 1
    for (int rt = 0; rt < EX_ROWS / 16; ++rt) {
        const unsigned char* ah_p = buf + (rt * 16 + n16) * ROWB + kq * 16;
        const unsigned char* al_p = ah_p + EX_STAGE_BYTES;
        f32x4 acc[EX_CT], acx[EX_CT];
#pragma unroll
        for (int ct = 0; ct < EX_CT; ++ct) { acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; acx[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ah_p + s * 64);
            const f16x8 al = *reinterpret_cast<const f16x8*>(al_p + s * 64);
#pragma unroll
            for (int ct = 0; ct < EX_CT; ++ct) {
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ct][s], acc[ct], 0, 0, 0);
                acx[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ct][s], acx[ct], 0, 0, 0);
                acx[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ct][s], acx[ct], 0, 0, 0);
            }
        }
        {
            const f16x4 ah = *reinterpret_cast<const f16x4*>(ah_p + 256 - kq * 8);
            const f16x4 al = *reinterpret_cast<const f16x4*>(al_p + 256 - kq * 8);
#pragma unroll
            for (int ct = 0; ct < EX_CT; ++ct) {
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bht[ct], acc[ct], 0, 0, 0);
                acx[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, blt[ct], acx[ct], 0, 0, 0);
                acx[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bht[ct], acx[ct], 0, 0, 0);
            }
        }
        // lane holds rows 4*kq + j (j = 0..3) of column n16: increasing j == increasing row, strict > keeps the first
        const int row0 = st * EX_ROWS + rt * 16 + 4 * kq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = row0 + j < n_ref;
#pragma unroll
            for (int ct = 0; ct < EX_CT; ++ct) {
                const float v = fmaf(acx[ct][j], LO_UNSCALE, acc[ct][j]);
                if (ok && v > best[ct]) { best[ct] = v; besti[ct] = row0 + j; }
            }
        }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void match_exact_kernel(
    const f16* __restrict__ lr_hi, const f16* __restrict__ lr_lo, const f16* __restrict__ ref_hi,
    const f16* __restrict__ ref_lo, int n_ref, const int32_t* __restrict__ flagged, unsigned long long* __restrict__ keys) {
    __shared__ __attribute__((aligned(16))) unsigned char lds0[2 * EX_STAGE_BYTES];     // stage buffers: [hi | lo][row][KP]
    __shared__ __attribute__((aligned(16))) unsigned char lds1[2 * EX_STAGE_BYTES];
    const int count = flagged[0];
    const int ngroups = (count + EX_COLS - 1) / EX_COLS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    const int n_stages = (n_ref + EX_ROWS - 1) / EX_ROWS;
    // work item = (column group, row part): the reference rows are split so that there are about as many items as
    // workgroups (the flagged count is only known here, on the device)
    const int nsplit = max(1, min(n_stages, (int)gridDim.x / max(ngroups, 1)));
    for (int item = blockIdx.x; item < ngroups * nsplit; item += gridDim.x) {
        const int g = item / nsplit, part = item - g * nsplit;
        const int s_begin = (int)((long long)n_stages * part / nsplit), s_end = (int)((long long)n_stages * (part + 1) / nsplit);
        __syncthreads();                                            // the previous item's readers are done with lds0
        ex_issue_stage(ref_hi, ref_lo, s_begin, lds0, wave, lane);
        // B operand: EX_CT tiles of 16 flagged columns per wave, straight from the LR patch rows (hi / lo): lane group kq
        // holds elements 32 s + 8 kq .. + 7 (s < 4) and 128 + 4 kq .. + 3 (tail) of its column
        f16x8 bh[EX_CT][4], bl[EX_CT][4];
        f16x4 bht[EX_CT], blt[EX_CT];
        int colv[EX_CT];
#pragma unroll
        for (int ct = 0; ct < EX_CT; ++ct) {                        // (all four list reads in flight before the row loads)
            const int fi = g * EX_COLS + (wave * EX_CT + ct) * 16 + n16;
            colv[ct] = flagged[1 + min(fi, count - 1)] | (fi < count ? 0 : 0x80000000);      // < 0: no column
        }
#pragma unroll
        for (int ct = 0; ct < EX_CT; ++ct) {
            const int col = colv[ct] & 0x7fffffff;
            const f16* bp0 = lr_hi + (size_t)col * KP + kq * 8;
            const f16* bp1 = lr_lo + (size_t)col * KP + kq * 8;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bh[ct][s] = *reinterpret_cast<const f16x8*>(bp0 + s * 32);
                bl[ct][s] = *reinterpret_cast<const f16x8*>(bp1 + s * 32);
            }
            bht[ct] = *reinterpret_cast<const f16x4*>(bp0 + 128 - kq * 4);
            blt[ct] = *reinterpret_cast<const f16x4*>(bp1 + 128 - kq * 4);
        }
        float best[EX_CT];
        int besti[EX_CT];
#pragma unroll
        for (int ct = 0; ct < EX_CT; ++ct) { best[ct] = -INFINITY; besti[ct] = 0x7fffffff; }
        // double-buffered stages, one barrier each: the barrier's vmcnt(0) lands stage st (issued one stage earlier, under
        // the previous stage's MFMAs) and frees the other buffer for stage st + 1
        for (int st = s_begin; st < s_end; st += 2) {
            __syncthreads();
            if (st + 1 < s_end) ex_issue_stage(ref_hi, ref_lo, st + 1, lds1, wave, lane);
            ex_compute_stage(lds0, st, n_ref, n16, kq, bh, bl, bht, blt, best, besti);
            if (st + 1 < s_end) {
                __syncthreads();
                if (st + 2 < s_end) ex_issue_stage(ref_hi, ref_lo, st + 2, lds0, wave, lane);
                ex_compute_stage(lds1, st + 1, n_ref, n16, kq, bh, bl, bht, blt, best, besti);
            }
        }
#pragma unroll
        for (int ct = 0; ct < EX_CT; ++ct) {
            unsigned long long k = besti[ct] == 0x7fffffff ? 0ull : ex_key(best[ct], besti[ct]);
            const unsigned long long k1 = __shfl_xor(k, 16);
            k = k1 > k ? k1 : k;
            const unsigned long long k2 = __shfl_xor(k, 32);
            k = k2 > k ? k2 : k;
            if (kq == 0 && colv[ct] >= 0 && k != 0ull) atomicMax(&keys[g * EX_COLS + (wave * EX_CT + ct) * 16 + n16], k);
        }
    }
}

__global__ __launch_bounds__(64) void match_exact_finish_kernel(const float* __restrict__ lf, int h, int w, const float* __restrict__ rf, int hr,
                                          int wr, const float* __restrict__ inv_lr, const float* __restrict__ inv_ref,
                                          const int32_t* __restrict__ flagged, const unsigned long long* __restrict__ keys,
                                          float* __restrict__ conf, int32_t* __restrict__ idx) {
    const int count = flagged[0];
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < count; f += gridDim.x * blockDim.x) {
        const unsigned long long k = keys[f];
        if (k == 0ull) continue;
        const int r = (int)(0xffffffffu - (unsigned)(k & 0xffffffffull));
        const int col = flagged[1 + f];
        const int y = col / w, x = col - y * w;
        int ly[3], lx[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) { ly[t] = rv_reflect(y + t - 1, h); lx[t] = rv_reflect(x + t - 1, w); }
        const float v = patch_dot<4>(lf, h, w, ly, lx, rf, hr, wr, r / wr, r % wr) * inv_lr[col] * inv_ref[r];
        const float c0 = conf[col];
        const int i0 = idx[col];
        if (v > c0 || (v == c0 && r < i0)) { conf[col] = v; idx[col] = r; }
    }
}

extern "C" int refvsr_match_exact(const float* lr_feat, int h, int w, const float* ref_feat, int hr, int wr,
                                  const void* lr_rows, const void* lr_rows_lo, const void* ref_rows, const void* ref_rows_lo,
                                  const float* inv_lr, const float* inv_ref, const int32_t* flagged, void* keys,
                                  float* conf, int32_t* idx, void* stream) {
    RV_CHECK(lr_feat && ref_feat && lr_rows && lr_rows_lo && ref_rows && ref_rows_lo && inv_lr && inv_ref && flagged &&
             keys && conf && idx, "match_exact: null pointer");
    RV_CHECK(h >= 2 && w >= 2 && hr >= 2 && wr >= 2, "match_exact: bad sizes");
    // fixed grid, two workgroups per CU (the flagged count lives on the device; the kernel sizes its work items from it)
    hipLaunchKernelGGL(match_exact_kernel, dim3(2 * rv_stream_cus((hipStream_t)stream)), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)lr_rows, (const f16*)lr_rows_lo, (const f16*)ref_rows, (const f16*)ref_rows_lo, hr * wr,
                       flagged, (unsigned long long*)keys);
    RV_LAUNCH_CHECK();
    // one wave per workgroup, as many workgroups as there can be flagged columns (the count lives on the device; idle ones exit
    // at once): the gathers of a wave are fully divergent, so the few thousand columns are spread over as many CUs as possible
    hipLaunchKernelGGL(match_exact_finish_kernel, dim3(rv_cdiv(h * w, 64)), dim3(64), 0, (hipStream_t)stream, lr_feat, h, w, ref_feat, hr,
                       wr, inv_lr, inv_ref, flagged, (const unsigned long long*)keys, conf, idx);
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
