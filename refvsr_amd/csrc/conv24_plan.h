// K plans of the compile-time-specialised 3x3 convolutions (conv24.hip, resblock48.hip): which K-block (tap, channel group) a
// lane quarter q consumes in K-step s, and the LDS byte offset of that block relative to a window origin.  The staged tile keeps
// one pixel in PS = ncg | 1 sixteen-byte slots (odd stride: bank-conflict-free B reads with the pixel permutation of common.h);
// XW = pixels per staged row (34 for a 1-pixel halo, 36 for the 2-pixel halo of a fused block).
#pragma once
#include <type_traits>
#include <utility>

namespace {
constexpr int C24_TW = 32, C24_XW = 34;                       // tile width; staged row of the single convs = 34 pixels

__host__ __device__ constexpr int c24_steps(int ncg) { return ncg == 1 ? 3 : ncg == 2 ? 5 : ncg == 3 ? 7 : ncg == 4 ? 9 : ncg == 6 ? 14 : ncg == 7 ? 18 : ncg == 12 ? 27 : 0; }
// K-block (K-step s, quarter q) -> ty << 16 | tx << 8 | cg, or -1 for a zero block
__host__ __device__ constexpr int c24_kblock(int ncg, int s, int q) {
    const int perm[4] = {0, 2, 1, 3};
    int ty = 0, tx = 0, cg = 0;
    if (ncg == 1) {
        if (q == 3) return -1;
        ty = s; tx = perm[q]; cg = 0;                                  // taps tx = 0, 2, 1 of row s (+ a zero block)
    } else if (ncg == 3) {
        if (s < 6) { const int u = 4 * (s & 1) + perm[q]; ty = s >> 1; tx = u / 3; cg = u % 3; }
        else { if (q == 3) return -1; ty = q; tx = 2; cg = 2; }
    } else if (ncg == 4) {
        ty = s / 3; tx = s % 3; cg = perm[q];
    } else if (ncg == 2) {
        if (s < 3) { const int u4[4] = {0, 4, 1, 3}; ty = s; tx = u4[q] / 3; cg = u4[q] % 3; }
        else if (s == 3) { ty = q & 1; tx = 2; cg = q >> 1; }
        else { if (q & 1) return -1; ty = 2; tx = 2; cg = q >> 1; }
    } else if (ncg == 7) {                                         // two steps per tap: cg = {0, 2, 1, 3}, then {4, 6, 5, zero block}
        ty = s / 6; tx = (s / 2) % 3;
        if (s & 1) { if (q == 3) return -1; cg = 4 + perm[q]; } else { cg = perm[q]; }
    } else if (ncg == 12) {                                        // three steps per tap: cg = 4 j + {0, 2, 1, 3}
        ty = s / 9; tx = (s / 3) % 3; cg = 4 * (s % 3) + perm[q];
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
__host__ __device__ constexpr int c24_off(int ncg, int s, int q, int xw = C24_XW) {
    int kb = c24_kblock(ncg, s, q);
    if (kb < 0) kb = c24_kblock(ncg, s, q - 1);
    const int ps = ncg | 1;
    return (kb >> 16) * (xw * ps * 16) + (((kb >> 8) & 255) * ps + (kb & 255)) * 16;
}
// pattern of a K-step: steps of one pattern differ only by an immediate
__host__ __device__ constexpr int c24_pat(int ncg, int s) {
    return ncg == 1 ? 0 : ncg == 3 ? (s < 6 ? 0 : 1) : ncg == 4 ? 0 : ncg == 12 ? 0 : ncg == 7 ? (s & 1) : ncg == 2 ? (s < 3 ? 0 : s - 2) : (s < 9 ? 0 : s < 12 ? 1 : s - 10);
}
__host__ __device__ constexpr int c24_npat(int ncg) { return ncg == 1 ? 1 : ncg == 3 ? 2 : ncg == 4 ? 1 : ncg == 12 ? 1 : ncg == 7 ? 2 : ncg == 2 ? 3 : 4; }
// first K-step of a pattern
__host__ __device__ constexpr int c24_pat_step(int ncg, int p) {
    return ncg == 1 ? 0 : ncg == 3 ? (p ? 6 : 0) : ncg == 4 ? 0 : ncg == 12 ? 0 : ncg == 7 ? p : ncg == 2 ? (p ? p + 2 : 0) : (p == 0 ? 0 : p == 1 ? 9 : p + 10);
}

// compile-time proof of the plans: every K-block of the 3 x 3 x ncg window exactly once, K-steps of one pattern differ by an
// immediate only, the quarters (0, 1) and (2, 3) of a step read slots of equal parity (bank-conflict-free ds_read_b128)
__host__ __device__ constexpr bool c24_plan_ok(int ncg, int xw = C24_XW) {
    const int S = c24_steps(ncg);
    int seen[3 * 3 * 16] = {};
    for (int s = 0; s < S; ++s) {
        const int ps = c24_pat_step(ncg, c24_pat(ncg, s));
        if (c24_pat(ncg, ps) != c24_pat(ncg, s)) return false;
        for (int q = 0; q < 4; ++q) {
            const int kb = c24_kblock(ncg, s, q);
            if (kb >= 0) {
                const int ty = kb >> 16, tx = (kb >> 8) & 255, cg = kb & 255;
                if (ty > 2 || tx > 2 || cg >= ncg) return false;
                seen[(ty * 3 + tx) * 16 + cg] += 1;
            } else if (q == 0) {
                return false;
            }
            if (c24_off(ncg, s, q, xw) - c24_off(ncg, s, 0, xw) != c24_off(ncg, ps, q, xw) - c24_off(ncg, ps, 0, xw)) return false;
        }
        if (((c24_off(ncg, s, 0, xw) ^ c24_off(ncg, s, 1, xw)) & 16) || ((c24_off(ncg, s, 2, xw) ^ c24_off(ncg, s, 3, xw)) & 16)) return false;
    }
    for (int t = 0; t < 9; ++t)
        for (int cg = 0; cg < ncg; ++cg)
            if (seen[t * 16 + cg] != 1) return false;
    for (int p = 0; p < c24_npat(ncg); ++p)
        if (c24_pat(ncg, c24_pat_step(ncg, p)) != p) return false;
    return true;
}
static_assert(c24_plan_ok(1) && c24_plan_ok(2) && c24_plan_ok(3) && c24_plan_ok(4) && c24_plan_ok(6) && c24_plan_ok(7) && c24_plan_ok(12), "conv24 K plan");

template <class F, int... I>
__device__ __forceinline__ void c24_static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
}  // namespace

