// Reference alignment kernels: index-driven block gather (AlignedAttention without `align`) and the
// affine-deformable bilinear patch sampler of AlignedConv2d.  Pure HBM/L2 gathers of 16-byte HWC
// channel groups; one lane per (output pixel, 8-channel group).
#include "common.h"

__global__ void block_gather_nhwc16_kernel(const f16* __restrict__ value, int hv, int wv, int cs,
                                           const int32_t* __restrict__ idx, int gh, int gw, int s,
                                           f16* __restrict__ out) {
    const int ng = cs / 8;
    const int ow = gw * s;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = blockIdx.y;
    if (i >= ow * ng) return;
    const int ox = i / ng;
    const int g = i - ox * ng;
    const int gy = oy / s, ky = oy - gy * s;
    const int gx = ox / s, kx = ox - gx * s;
    const int wr = wv / s;
    const int id = idx[gy * gw + gx];
    const int ry = id / wr, rx = id - ry * wr;
    const int sy = min(ry * s + ky, hv - 1);     // indices are in-range by construction; clamp = memory safety
    const int sx = min(rx * s + kx, wv - 1);
    const uint4 v = *reinterpret_cast<const uint4*>(value + ((size_t)sy * wv + sx) * cs + g * 8);
    *reinterpret_cast<uint4*>(out + ((size_t)oy * ow + ox) * cs + g * 8) = v;
}

extern "C" int refvsr_block_gather_nhwc16(const void* value, int hv, int wv, int cs, const int32_t* idx, int gh, int gw,
                                          int s, void* out, void* stream) {
    RV_CHECK(value && idx && out && hv > 0 && wv > 0 && cs % 8 == 0 && gh > 0 && gw > 0 && s >= 1 && wv / s > 0,
             "block_gather_nhwc16: bad args");
    const int ng = cs / 8;
    hipLaunchKernelGGL(block_gather_nhwc16_kernel, dim3(rv_cdiv(gw * s * ng, 256), gh * s), dim3(256), 0,
                       (hipStream_t)stream, (const f16*)value, hv, wv, cs, idx, gh, gw, s, (f16*)out);
    RV_LAUNCH_CHECK();
    return 0;
}

__global__ void block_gather_rgb_kernel(const float* __restrict__ value, int hv, int wv, const int32_t* __restrict__ idx,
                                        int gh, int gw, int s, f16* __restrict__ out8, float* __restrict__ out_planar) {
    const int ow = gw * s;
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = blockIdx.y;
    if (ox >= ow) return;
    const int gy = oy / s, ky = oy - gy * s;
    const int gx = ox / s, kx = ox - gx * s;
    const int wr = wv / s;
    const int id = idx[gy * gw + gx];
    const int ry = id / wr, rx = id - ry * wr;
    const int sy = min(ry * s + ky, hv - 1);
    const int sx = min(rx * s + kx, wv - 1);
    const size_t plane = (size_t)hv * wv;
    const size_t sp = (size_t)sy * wv + sx;
    const float r = value[sp], g = value[plane + sp], b = value[2 * plane + sp];
    if (out8) {
        f16x8 o = {(f16)r, (f16)g, (f16)b, 0, 0, 0, 0, 0};
        *reinterpret_cast<f16x8*>(out8 + ((size_t)oy * ow + ox) * 8) = o;
    }
    if (out_planar) {                                   // exact copy (the `vis` debugging samples, RefVSR.py:305-309)
        const size_t op = (size_t)gh * s * ow, o = (size_t)oy * ow + ox;
        out_planar[o] = r; out_planar[op + o] = g; out_planar[2 * op + o] = b;
    }
}

extern "C" int refvsr_block_gather_rgb(const float* value, int hv, int wv, const int32_t* idx, int gh, int gw, int s,
                                       void* out8, float* out_planar, void* stream) {
    RV_CHECK(value && idx && (out8 || out_planar) && hv > 0 && wv > 0 && gh > 0 && gw > 0 && s >= 1 && wv / s > 0,
             "block_gather_rgb: bad args");
    hipLaunchKernelGGL(block_gather_rgb_kernel, dim3(rv_cdiv(gw * s, 128), gh * s), dim3(128), 0, (hipStream_t)stream,
                       value, hv, wv, idx, gh, gw, s, (f16*)out8, out_planar);
    RV_LAUNCH_CHECK();
    return 0;
}

// AlignedConv2d sampler (RefVSR_/alignment.py:53-100,102-178).  "x" of the reference
// is the ROW axis.  Output pixel (ks*i + a, ks*j + b) samples the reflection-padded(1) map at
//   pr = (off_a*s_x)*cos + (off_b*s_y)*(-sin) + half + 0.5 + (1 + ks*i)
//   pc = (off_a*s_x)*sin + (off_b*s_y)*cos    + half + 0.5 + (1 + ks*j)
// with floor/ceil corners and the coordinate itself clamped to the padded extent and bilinear
// weights formed from the clamped values.
__global__ void aligned_sample_kernel(const f16* __restrict__ x, int h, int w, int ks, int cs,
                                      const float* __restrict__ affine, f16* __restrict__ out) {
    const int ng = cs / 8;
    const int W2 = w * ks, H2 = h * ks;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int Y = blockIdx.y;
    if (i >= W2 * ng) return;
    const int X = i / ng;
    const int g = i - X * ng;
    const int li = Y / ks, a = Y - li * ks;
    const int lj = X / ks, b = X - lj * ks;
    const size_t ap = (size_t)li * w + lj;
    const size_t aplane = (size_t)h * w;
    const float s_x = affine[ap];
    const float s_y = affine[aplane + ap];
    const float th = (affine[2 * aplane + ap] - 1.0f) * 1.0472f;
    const int half = (ks - 1) / 2;
    const float off_a = (float)a - (float)half - 0.5f;
    const float off_b = (float)b - (float)half - 0.5f;
    const float px = off_a * s_x;
    const float py = off_b * s_y;
    const float cs_ = cosf(th), sn = sinf(th);
    const float rx = px * cs_ + py * (-sn);
    const float ry = px * sn + py * cs_;
    const float hp1 = (float)(H2 + 1), wp1 = (float)(W2 + 1);     // padded extent - 1
    float pr = ((rx + (float)half) + 0.5f) + (float)(1 + ks * li);
    float pc = ((ry + (float)half) + 0.5f) + (float)(1 + ks * lj);
    float r0 = floorf(pr), c0 = floorf(pc);
    float r1 = fminf(fmaxf(r0 + 1.0f, 0.0f), hp1), c1 = fminf(fmaxf(c0 + 1.0f, 0.0f), wp1);
    r0 = fminf(fmaxf(r0, 0.0f), hp1);
    c0 = fminf(fmaxf(c0, 0.0f), wp1);
    pr = fminf(fmaxf(pr, 0.0f), hp1);
    pc = fminf(fmaxf(pc, 0.0f), wp1);
    const float g_lt = (1.0f + (r0 - pr)) * (1.0f + (c0 - pc));
    const float g_rb = (1.0f - (r1 - pr)) * (1.0f - (c1 - pc));
    const float g_lb = (1.0f + (r0 - pr)) * (1.0f - (c1 - pc));
    const float g_rt = (1.0f - (r1 - pr)) * (1.0f + (c0 - pc));
    // padded index -> source index through the reflection pad
    const int sr0 = rv_reflect((int)r0 - 1, H2), sr1 = rv_reflect((int)r1 - 1, H2);
    const int sc0 = rv_reflect((int)c0 - 1, W2), sc1 = rv_reflect((int)c1 - 1, W2);
    const f16x8 v_lt = *reinterpret_cast<const f16x8*>(x + ((size_t)sr0 * W2 + sc0) * cs + g * 8);
    const f16x8 v_rb = *reinterpret_cast<const f16x8*>(x + ((size_t)sr1 * W2 + sc1) * cs + g * 8);
    const f16x8 v_lb = *reinterpret_cast<const f16x8*>(x + ((size_t)sr0 * W2 + sc1) * cs + g * 8);
    const f16x8 v_rt = *reinterpret_cast<const f16x8*>(x + ((size_t)sr1 * W2 + sc0) * cs + g * 8);
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        o[k] = (f16)(g_lt * (float)v_lt[k] + g_rb * (float)v_rb[k] + g_lb * (float)v_lb[k] + g_rt * (float)v_rt[k]);
    *reinterpret_cast<f16x8*>(out + ((size_t)Y * W2 + X) * cs + g * 8) = o;
}

extern "C" int refvsr_aligned_sample(const void* x, int h, int w, int ks, int cs, const float* affine, void* out,
                                     void* stream) {
    RV_CHECK(x && affine && out && h > 0 && w > 0 && ks >= 1 && ks <= 16 && cs % 8 == 0, "aligned_sample: bad args");
    RV_CHECK(h * ks >= 2 && w * ks >= 2, "aligned_sample: map too small for reflection padding");
    const int ng = cs / 8;
    hipLaunchKernelGGL(aligned_sample_kernel, dim3(rv_cdiv(w * ks * ng, 256), h * ks), dim3(256), 0,
                       (hipStream_t)stream, (const f16*)x, h, w, ks, cs, affine, (f16*)out);
    RV_LAUNCH_CHECK();
    return 0;
}
