// Fused cosine GEMM + column top-2 of the reference matching (see match.hip for the operand layout and the exactness
// argument).  Its own translation unit because it is compiled with -fno-honor-nans (Makefile): the scores are finite by
// construction (finite fp16 operands, zero pads), and without NaN semantics the 16-value column maxima become 8
// v_max3_f32 per accumulator instead of 16 canonicalisations + 15 v_max_f32 -- the vector ALU port, which MFMA issue
// shares, is what bounds this kernel (PMC: 8.2 VALU instructions per MFMA before).
#include <type_traits>

#include "match_common.h"

// ---------------------------------------------------------------------------------------------
// fused GEMM + column top-2
// ---------------------------------------------------------------------------------------------
struct Top2 { float m1, m2; int i1, i2; };

__device__ __forceinline__ bool rv_better(float va, int ia, float vb, int ib) {
    return va > vb || (va == vb && ia < ib);
}

// A lane's 16 accumulator values are four quads of consecutive rows (register r <-> row 8 (r >> 2) + (r & 3)).  q[g] = max of
// quad g, return value = max of all 16: 10 instructions (v_max3_f32 + v_max_f32 per quad; no NaN semantics in this file).
__device__ __forceinline__ float acc_max(const f32x16& a, float (&q)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) q[g] = fmaxf(fmaxf(fmaxf(a[4 * g], a[4 * g + 1]), a[4 * g + 2]), a[4 * g + 3]);
    return fmaxf(fmaxf(fmaxf(q[0], q[1]), q[2]), q[3]);
}

// In-order insertion into the lane's running top-2 (increasing r == increasing row index, strict compares: the first of
// equal values wins), quad by quad: only quads whose maximum can still enter the top-2 are walked -- late in the row
// stream a wave takes the slow path for ONE value of ONE lane, and this makes that cost 4 x 8 instead of 16 x 8
// instructions.  Branch-free selects inside a quad.  MASK: rows >= n_ref (zero pad rows of the last stage) are excluded.
template <bool MASK>
__device__ __forceinline__ void top2_insert(Top2& s, const f32x16& acc, const float (&q)[4], float thr, int rowbase, int n_ref) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (q[g] > s.m2 && q[g] >= thr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = rowbase + j + 8 * g;
                const float v = (MASK && row >= n_ref) ? -INFINITY : acc[4 * g + j];
                const bool g1 = v > s.m1;
                const bool g2 = v > s.m2;
                s.m2 = g1 ? s.m1 : (g2 ? v : s.m2);
                s.i2 = g1 ? s.i1 : (g2 ? row : s.i2);
                s.m1 = g1 ? v : s.m1;
                s.i1 = g1 ? row : s.i1;
            }
        }
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
    float thr[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) { st[ct].m1 = st[ct].m2 = -INFINITY; st[ct].i1 = st[ct].i2 = 0; thr[ct] = -INFINITY; }

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
        auto run_pairs = [&](auto mask_c, const int rp_begin, const int rp_end) {
        constexpr bool MASK = decltype(mask_c)::value;
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
                float q0[4], q1[4];
                const float t0 = acc_max(accA0, q0), t1 = acc_max(accA1, q1);
                const bool in0 = t0 > st[0].m2 && t0 >= thr[0], in1 = t1 > st[1].m2 && t1 >= thr[1];
                if (in0 || in1) {
                    const int rowbase = c * CHUNK4 + (2 * rp) * 32 + 4 * hi;
                    if (in0) top2_insert<MASK>(st[0], accA0, q0, thr[0], rowbase, n_ref);
                    if (in1) top2_insert<MASK>(st[1], accA1, q1, thr[1], rowbase, n_ref);
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
                float q0[4], q1[4];
                const float t0 = acc_max(accB0, q0), t1 = acc_max(accB1, q1);
                const bool in0 = t0 > st[0].m2 && t0 >= thr[0], in1 = t1 > st[1].m2 && t1 >= thr[1];
                if (in0 || in1) {
                    const int rowbase = c * CHUNK4 + (2 * rp + 1) * 32 + 4 * hi;
                    if (in0) top2_insert<MASK>(st[0], accB0, q0, thr[0], rowbase, n_ref);
                    if (in1) top2_insert<MASK>(st[1], accB1, q1, thr[1], rowbase, n_ref);
                }
            }
        }
        };
        // A value can only enter the column's final top-2 if it also reaches the runner-up of the PARTNER lane (lanes l and
        // l ^ 32 share a column and see disjoint rows); >= because an equal value from an earlier row wins the merge.
        // The snapshot is refreshed once per stage (stale = smaller = still a superset).
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) thr[ct] = __shfl_xor(st[ct].m2, 32);
        const bool tail = (c + 1) * CHUNK4 > n_ref;       // only the last stage holds pad rows
        if (tail) run_pairs(std::true_type{}, 0, NRT / 4); else run_pairs(std::false_type{}, 0, NRT / 4);
        asm volatile("" ::: "memory");        // mid-stage: park the first half of the next stage, fetch the second
        if (has_next) {
            PF_STORE(lds[buf ^ 1], 0);
            PF_LOAD((size_t)(c + 1) * CHUNK4_U4 + HALF_U4);
        }
        asm volatile("" ::: "memory");
        if (tail) run_pairs(std::true_type{}, NRT / 4, NRT / 2); else run_pairs(std::false_type{}, NRT / 4, NRT / 2);
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

