// Shared helpers for the gfx950 RefVSR kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/refvsr_hip.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void refvsr_set_error(const char* fmt, ...);

#define RV_CHECK(cond, ...)                        \
    do {                                           \
        if (!(cond)) {                             \
            refvsr_set_error(__VA_ARGS__);         \
            return 1;                              \
        }                                          \
    } while (0)

#define RV_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            refvsr_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

#define RV_LAUNCH_CHECK() RV_HIP(hipGetLastError())

static inline int rv_cdiv(int a, int b) { return (a + b - 1) / b; }

// Per-device caches (function attributes, occupancy, CU count): one process may drive several GPUs.
#define RV_MAX_DEVICES 16
static inline int rv_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= RV_MAX_DEVICES) d = 0;
    return d;
}
static inline int rv_num_cus() {
    static int n_cu[RV_MAX_DEVICES] = {};
    const int d = rv_device();
    if (n_cu[d] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess) n_cu[d] = prop.multiProcessorCount;
        if (n_cu[d] <= 0) n_cu[d] = 256;
    }
    return n_cu[d];
}

// CUs the persistent launchers may count on for a launch on `st`: the budget registered for a CU-masked stream
// (refvsr_stream_create_cu_range / refvsr_stream_set_cu_budget, runtime.hip), otherwise the whole device.
int rv_stream_cus(hipStream_t st);

// Persistent tile walk, XCD-aware and balanced.  Workgroup b runs on XCD b % 8 (observed placement, used for speed only):
// the workgroups of one XCD get consecutive ranks, rank r walks the CONTIGUOUS tile range [r*n/g, (r+1)*n/g), so every
// workgroup has floor or ceil(n/g) tiles (a strided walk inside fixed eighths of the frame left single workgroups with
// twice the tiles of the others whenever g ~ n) and the tiles in flight on one XCD are neighbours sharing halo rows in
// that XCD's L2.
// (g = gridDim.x, passed in the kernel's own argument struct: read from the hidden dispatch arguments it costs a
// separate scalar-load round trip)
__device__ __forceinline__ void rv_tile_range(const int n_tiles, const int g, int& t_begin, int& t_end) {
    const int b = (int)blockIdx.x;
    int rank = b;
    if (g >= 8) {
        const int xcd = b & 7;
        rank = b >> 3;
        for (int y = 0; y < xcd; ++y) rank += (g - y + 7) >> 3;
    }
    // 32-bit unsigned divisions (~30 instructions each; the 64-bit ones hipcc expands to several hundred instructions
    // PER WAVE, at the head of every persistent launch: an s_memtime probe put ~4 000 cycles between kernel entry and the
    // first x-tile load).  rank * n_tiles < 2^31: at most 1 024 workgroups x 2 M tiles -- checked by the callers' hosts.
    const unsigned un = (unsigned)n_tiles, ug = (unsigned)g;
    t_begin = (int)(((unsigned)rank * un) / ug);
    t_end = (int)(((unsigned)(rank + 1) * un) / ug);
}

// ReflectionPad2d index (no edge repeat): -1 -> 1, n -> n-2.  Valid for -n < i < 2n-1.
__device__ __forceinline__ int rv_reflect(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

__device__ __forceinline__ float rv_lrelu(float v, float slope) { return v >= 0.f ? v : v * slope; }

// torch.linspace(-1, 1, n)[j] as computed by ATen's CPU kernel (symmetric two-sided evaluation).
__device__ __forceinline__ float rv_linspace_m1p1(int j, int n) {
    const float step = 2.0f / (float)(n - 1);
    return (j < n / 2) ? (-1.0f + step * (float)j) : (1.0f - step * (float)(n - 1 - j));
}

// warp (models/utils.py:35-43): zeros padding, align_corners=False sampling of a linspace(-1,1) grid displaced by the flow.
// Shared by the stand-alone warp kernels (resample.hip) and the conv kernels that warp a source while staging it.
struct WarpCoord { int x0, y0; float w00, w01, w10, w11; bool v00, v01, v10, v11; };

__device__ __forceinline__ WarpCoord warp_coord_uv(const float u, const float v, int hf, int wf, int hin, int win, int y, int x);
__device__ __forceinline__ WarpCoord warp_coord(const float* flow, int hf, int wf, int hin, int win, int y, int x) {
    const size_t fp = (size_t)y * wf + x;
    return warp_coord_uv(flow[fp], flow[(size_t)hf * wf + fp], hf, wf, hin, win, y, x);
}
// (u, v): the flow at grid pixel (y, x)
__device__ __forceinline__ WarpCoord warp_coord_uv(const float u, const float v, int hf, int wf, int hin, int win, int y, int x) {
    const float gx = rv_linspace_m1p1(x, wf) + u / (((float)win - 1.0f) / 2.0f);
    const float gy = rv_linspace_m1p1(y, hf) + v / (((float)hin - 1.0f) / 2.0f);
    const float xs = ((gx + 1.0f) * (float)win - 1.0f) / 2.0f;
    const float ys = ((gy + 1.0f) * (float)hin - 1.0f) / 2.0f;
    const float fx = floorf(xs), fy = floorf(ys);
    const float tx = xs - fx, ty = ys - fy;
    WarpCoord c;
    // clamp before the int conversion so wild flows cannot overflow; such taps are out of range anyway
    c.x0 = (int)fminf(fmaxf(fx, -2.0f), (float)win + 1.0f);
    c.y0 = (int)fminf(fmaxf(fy, -2.0f), (float)hin + 1.0f);
    const bool xin0 = c.x0 >= 0 && c.x0 < win, xin1 = c.x0 + 1 >= 0 && c.x0 + 1 < win;
    const bool yin0 = c.y0 >= 0 && c.y0 < hin, yin1 = c.y0 + 1 >= 0 && c.y0 + 1 < hin;
    c.v00 = xin0 && yin0; c.v01 = xin1 && yin0; c.v10 = xin0 && yin1; c.v11 = xin1 && yin1;
    c.w00 = (1.0f - ty) * (1.0f - tx); c.w01 = (1.0f - ty) * tx;
    c.w10 = ty * (1.0f - tx); c.w11 = ty * tx;
    return c;
}

// one 16-byte channel group of warp(x, flow) at grid pixel (y, px): fp32 blend in tap order 00, 01, 10, 11, then fp16
__device__ __forceinline__ uint4 warp_group16(const unsigned char* x, int pixb, int hin, int win, const WarpCoord& c, int goff) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto tap = [&](bool valid, int yy, int xx, float wgt) {
        if (valid) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(x + ((size_t)yy * win + xx) * pixb + goff);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += wgt * (float)v[k];
        }
    };
    tap(c.v00, c.y0, c.x0, c.w00);
    tap(c.v01, c.y0, c.x0 + 1, c.w01);
    tap(c.v10, c.y0 + 1, c.x0, c.w10);
    tap(c.v11, c.y0 + 1, c.x0 + 1, c.w11);
    union { f16x8 h; uint4 u; } o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.h[k] = (f16)acc[k];
    return o.u;
}

// align_corners=True bilinear taps of output index o (n_in -> n_out samples): shared by resample.hip's resize_kernel and by the
// kernels that evaluate the up-sampled map in place.  Contraction is switched OFF here: under hipcc's default -ffp-contract=fast
// `x - (float)i0` may or may not fuse with the product that made x, depending on the code around the inlined copy -- two
// kernels restating the same taps then differ by an ulp (found on the GPU: refvsr_warp_nhwc16_up2 vs resize + warp).
__device__ __forceinline__ void rv_bilinear_ac_src(int o, int n_in, int n_out, int& i0, int& i1, float& l1) {
#pragma clang fp contract(off)
    const float sc = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.0f;
    const float x = (float)o * sc;
    i0 = min((int)x, n_in - 1);
    i1 = min(i0 + 1, n_in - 1);
    l1 = x - (float)i0;
}
// F.interpolate(s, scale_factor=2, mode='bilinear', align_corners=True) of a planar fp32 map [h][w] at output pixel (oy, ox)
// of the [2h][2w] grid: the taps and FMA chains of resample.hip's resize_kernel<BILINEAR_AC> (flow_up2 of the engine)
__device__ __forceinline__ float rv_bilinear_ac2_at(const float* __restrict__ s, int h, int w, int oy, int ox) {
    int y0, y1, x0, x1;
    float ly, lx;
    rv_bilinear_ac_src(oy, h, 2 * h, y0, y1, ly);
    rv_bilinear_ac_src(ox, w, 2 * w, x0, x1, lx);
    const float wy0 = 1.0f - ly, wx0 = 1.0f - lx;
    const float r0 = fmaf(lx, s[(size_t)y0 * w + x1], fmaf(wx0, s[(size_t)y0 * w + x0], 0.0f));
    const float r1 = fmaf(lx, s[(size_t)y1 * w + x1], fmaf(wx0, s[(size_t)y1 * w + x0], 0.0f));
    return fmaf(ly, r1, fmaf(wy0, r0, 0.0f));
}

// ---- bicubic F.interpolate taps (ATen upsample_bicubic2d, A = -0.75, align_corners=False): shared by resample.hip's
// resize_kernel and the kernels that evaluate the up-sampled map on the fly (conv24.hip: CONF variant)
__device__ __forceinline__ void rv_cubic_taps(float t, float* wgt) {
    const float A = -0.75f;
    float x = t + 1.0f;
    wgt[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
    x = t;
    wgt[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 1.0f - t;
    wgt[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 2.0f - t;
    wgt[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}
// source indices (clamped) and weights of output index o; scale = source step per output sample
__device__ __forceinline__ void rv_cubic_src(int o, int n_in, float scale, int* idx, float* wgt) {
    const float x = ((float)o + 0.5f) * scale - 0.5f;
    const float fl = floorf(x);
    const int ix = (int)fl;
    rv_cubic_taps(x - fl, wgt);
#pragma unroll
    for (int k = 0; k < 4; ++k) idx[k] = min(max(ix - 1 + k, 0), n_in - 1);
}
// one bicubic sample of a planar fp32 map [h][w] at output pixel (oy, ox): the FMA chains of resize_kernel<BICUBIC>
__device__ __forceinline__ float rv_bicubic_at(const float* __restrict__ s, int h, int w, int oy, int ox, float sy, float sx) {
    int iy[4], ix[4];
    float wy[4], wx[4];
    rv_cubic_src(oy, h, sy, iy, wy);
    rv_cubic_src(ox, w, sx, ix, wx);
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r = 0.0f;
        const float* row = s + (size_t)iy[j] * w;
#pragma unroll
        for (int i = 0; i < 4; ++i) r = fmaf(wx[i], row[ix[i]], r);
        acc = fmaf(wy[j], r, acc);
    }
    return acc;
}

// ---- result formats of the output head (ABI 14, REFVSR_RESULT_*): one value v in [0, 1] of the planar [3][h][w] result at element e.
// U8 = what the reference's consumers make of the fp32 frame on the CPU (evaluation/eval_qual_quan.py:117-119: cv2.imwrite of
// output * 255 = saturate_cast<uchar>, round to nearest even): rint(v * 255) in fp32, the same two roundings.
__device__ __forceinline__ void rv_store_result(void* out, const size_t e, const float v, const int fmt) {
    if (fmt == REFVSR_RESULT_F32) reinterpret_cast<float*>(out)[e] = v;
    else if (fmt == REFVSR_RESULT_F16) reinterpret_cast<f16*>(out)[e] = (f16)v;
    else reinterpret_cast<unsigned char*>(out)[e] = (unsigned char)__float2int_rn(v * 255.0f);
}

// ---- K-block order of the MFMA convolutions (shared with refvsr_amd/packing.py:kslot) --------------------------
// A K-block is one 16-byte channel group `cg` of one tap (ty, tx).  A wave's ds_read_b128 of the B operand is
// served in four groups of 16 lanes, each mixing TWO adjacent K-blocks (q = 0|1 or 2|3, MI355X_MICROARCH.md
// "LDS": {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...).  With the pixel -> lane permutation rv_pix16 the eight lanes
// of one K-block in a group sit on even (or odd) pixels, so a group is bank-conflict free iff the two K-blocks'
// LDS slot offsets have the same parity = (tx + cg) & 1 (tile pitch even, pixel stride odd).  K-blocks are
// therefore ordered: all even-parity blocks in natural (ty, tx, cg) order, one zero block if their count E is odd,
// all odd-parity blocks, zero blocks up to a multiple of 4.
__host__ __device__ inline int rv_keven(int ks, int ncg) {
    const int ce = (ncg + 1) >> 1, co = ncg >> 1;
    return ks * (((ks + 1) >> 1) * ce + (ks >> 1) * co);
}
__host__ __device__ inline int rv_ksteps(int ks, int ncg) { return (ks * ks * ncg + (rv_keven(ks, ncg) & 1) + 3) / 4; }
__host__ __device__ inline int rv_kslot(int ty, int tx, int cg, int ks, int ncg) {
    const int ce = (ncg + 1) >> 1, co = ncg >> 1;
    const int p = (tx + cg) & 1;
    const int c0 = p ? co : ce, c1 = p ? ce : co;                 // class-p blocks per tap with even | odd tx
    const int row = ((ks + 1) >> 1) * c0 + (ks >> 1) * c1;
    const int rank = ty * row + ((tx + 1) >> 1) * c0 + (tx >> 1) * c1 + (cg >> 1);
    if (!p) return rank;
    const int E = rv_keven(ks, ncg);
    return E + (E & 1) + rank;
}
// j-th zero block (j = 0 .. 4*S - G - 1) -> slot
__host__ __device__ inline int rv_kpad_slot(int j, int ks, int ncg) {
    const int E = rv_keven(ks, ncg);
    return ((E & 1) && j == 0) ? E : ks * ks * ncg + j;
}
// MFMA column n (= lane & 15) -> pixel of the 16-pixel tile: the lanes a ds_read_b128 group takes from one K-block
// ({0-3, 12-15} | {4-11}) land on the even | odd pixels.
__device__ __forceinline__ int rv_pix16(int n) { return (n < 4 || n >= 12) ? ((n & 7) << 1) : (((n - 4) << 1) | 1); }
