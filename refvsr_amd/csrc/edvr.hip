// Kernels of the RefVSR_IR variant's EDVR-M feature extractor (models/archs/RefVSR_IR.py:424-546, edvr_net.py) that
// are not convolutions: the sampling half of the modulated deformable convolution (DCNv2), the temporal-attention
// weighting of TSAFusion, 3x3 stride-2 pooling, x2 bilinear upsampling and the attention blend -- all on fp16 HWC maps.
// The convolutions themselves (incl. the DCN's contraction, a 1x1 conv over the sampled columns) run on conv_mfma.hip.
// HBM-bound gathers / elementwise passes: one lane per 16-byte channel group.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------------
// Modulated deformable convolution, sampling half (mmcv modulated_deform_conv: dmcn_im2col_bilinear; edvr_net.py:49-56).
// For output pixel (y, x), tap k = ky*3 + kx, deformable group g (8 channels = one 16-byte group):
//   cols[y][x][k*C + g*8 .. +8] = sigmoid(om[2*DG*9 + g*9 + k]) * bilinear(x_g, y + ky - 1 + om[g*18 + 2k], x + kx - 1 + om[g*18 + 2k + 1])
// samples outside (-1, H) x (-1, W) are 0, corner pixels outside the map contribute 0.  om: planar fp32 [3*DG*9][h][w]
// (the raw conv_offset output: [o1 | o2 | mask], cat(o1, o2) being the offset tensor).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void dcn_sample_kernel(const f16* __restrict__ x, int h, int w, int c, const float* __restrict__ om, int dg,
                                  f16* __restrict__ cols) {
    const int per_px = 9 * dg;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)h * w * per_px;
    if (i >= total) return;
    const int pix = (int)(i / per_px);
    const int r = (int)(i - (long long)pix * per_px);
    const int k = r / dg, g = r - k * dg;
    const int y = pix / w, xx = pix - y * w;
    const size_t plane = (size_t)h * w;
    const float oy = om[(size_t)(g * 18 + 2 * k) * plane + pix];
    const float ox = om[(size_t)(g * 18 + 2 * k + 1) * plane + pix];
    const float mr = om[(size_t)(2 * dg * 9 + g * 9 + k) * plane + pix];
    const float m = 1.0f / (1.0f + __expf(-mr));
    const float py = (float)(y + k / 3 - 1) + oy, px = (float)(xx + k % 3 - 1) + ox;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (py > -1.0f && px > -1.0f && py < (float)h && px < (float)w) {
        const float fy = floorf(py), fx = floorf(px);
        const float ly = py - fy, lx = px - fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const float wgt[4] = {(1.0f - ly) * (1.0f - lx), (1.0f - ly) * lx, ly * (1.0f - lx), ly * lx};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int yy = y0 + (t >> 1), xc = x0 + (t & 1);
            if (yy >= 0 && yy < h && xc >= 0 && xc < w) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(x + ((size_t)yy * w + xc) * c + g * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += wgt[t] * (float)v[j];
            }
        }
    }
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)(acc[j] * m);
    *reinterpret_cast<f16x8*>(cols + (size_t)pix * 9 * c + (size_t)k * c + g * 8) = o;
}

extern "C" int refvsr_dcn_sample(const void* x, int h, int w, int c, const float* offset_mask, int deform_groups,
                                 void* cols, void* stream) {
    RV_CHECK(x && offset_mask && cols && h > 0 && w > 0 && c > 0 && deform_groups > 0 && c == deform_groups * 8,
             "dcn_sample: bad args (needs 8 channels per deformable group, c=%d groups=%d)", c, deform_groups);
    const long long total = (long long)h * w * 9 * deform_groups;
    hipLaunchKernelGGL(dcn_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)x, h, w, c, offset_mask, deform_groups, (f16*)cols);
    RV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// TSAFusion, temporal attention (edvr_net.py:259-272): corr_i = sigmoid(sum_c emb_i * emb_ref); out[.., i*C + c] =
// aligned_i[.., c] * corr_i -- the t weighted maps written side by side (the input of the two 1x1 fusion convs).
// ---------------------------------------------------------------------------------------------------------------------
struct TsaArgs { const f16* aligned[8]; const f16* emb[8]; const f16* emb_ref; f16* out; int t, c, npix; };

__global__ void tsa_weight_kernel(TsaArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.npix * a.t) return;
    const int pix = i / a.t, f = i - pix * a.t;
    const f16* e = a.emb[f] + (size_t)pix * a.c;
    const f16* r = a.emb_ref + (size_t)pix * a.c;
    float dot = 0.0f;
    for (int g = 0; g < a.c; g += 8) {
        const f16x8 ev = *reinterpret_cast<const f16x8*>(e + g), rv = *reinterpret_cast<const f16x8*>(r + g);
#pragma unroll
        for (int j = 0; j < 8; ++j) dot += (float)ev[j] * (float)rv[j];
    }
    const float p = 1.0f / (1.0f + __expf(-dot));
    const f16* s = a.aligned[f] + (size_t)pix * a.c;
    f16* d = a.out + (size_t)pix * a.t * a.c + (size_t)f * a.c;
    for (int g = 0; g < a.c; g += 8) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(s + g);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)v[j] * p);
        *reinterpret_cast<f16x8*>(d + g) = o;
    }
}

extern "C" int refvsr_tsa_weight(const void* const* aligned, const void* const* emb, const void* emb_ref, int t, int c,
                                 int npix, void* out, void* stream) {
    RV_CHECK(aligned && emb && emb_ref && out && t >= 1 && t <= 8 && c % 8 == 0 && npix > 0, "tsa_weight: bad args");
    TsaArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < t; ++i) {
        RV_CHECK(aligned[i] && emb[i], "tsa_weight: null map %d", i);
        a.aligned[i] = (const f16*)aligned[i];
        a.emb[i] = (const f16*)emb[i];
    }
    a.emb_ref = (const f16*)emb_ref; a.out = (f16*)out; a.t = t; a.c = c; a.npix = npix;
    hipLaunchKernelGGL(tsa_weight_kernel, dim3(rv_cdiv(npix * t, 256)), dim3(256), 0, (hipStream_t)stream, a);
    RV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// MaxPool2d / AvgPool2d (3, stride 2, padding 1; the average counts the padding) on an fp16 HWC map, written into channels
// [c_off, c_off + c) of an output with channel stride out_c (edvr_net.py:214-215,277-279: cat([max, avg])).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void pool3s2_kernel(const f16* __restrict__ x, int h, int w, int c, f16* __restrict__ out, int ho, int wo,
                               int out_c, int c_off, int is_max) {
    const int ng = c / 8;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ho * wo * ng) return;
    const int pix = i / ng, g = i - pix * ng;
    const int oy = pix / wo, ox = pix - oy * wo;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = is_max ? -INFINITY : 0.0f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int yy = 2 * oy - 1 + dy, xx = 2 * ox - 1 + dx;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(x + ((size_t)yy * w + xx) * c + g * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = is_max ? fmaxf(acc[j], (float)v[j]) : acc[j] + (float)v[j];
            }
        }
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)(is_max ? acc[j] : acc[j] * (1.0f / 9.0f));
    *reinterpret_cast<f16x8*>(out + (size_t)pix * out_c + c_off + g * 8) = o;
}

extern "C" int refvsr_pool3s2_nhwc16(const void* x, int h, int w, int c, void* out, int out_c, int c_off, int is_max,
                                     void* stream) {
    RV_CHECK(x && out && h > 0 && w > 0 && c % 8 == 0 && out_c % 8 == 0 && c_off % 8 == 0 && c_off + c <= out_c,
             "pool3s2: bad args");
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    hipLaunchKernelGGL(pool3s2_kernel, dim3(rv_cdiv(ho * wo * (c / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)x, h, w, c, (f16*)out, ho, wo, out_c, c_off, is_max);
    RV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// nn.Upsample(scale_factor=2, bilinear, align_corners=False) of an fp16 HWC map, times `mul` (edvr_net.py:131,179-181:
// offsets are doubled when upsampled; :246).  Source coordinate (o + 0.5)/2 - 0.5 clamped at 0, upper neighbour clamped.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void up2_bilinear_kernel(const f16* __restrict__ x, int h, int w, int c, float mul, f16* __restrict__ out) {
    const int ng = c / 8;
    const int W2 = 2 * w;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * h * w * ng) return;
    const int pix = i / ng, g = i - pix * ng;
    const int oy = pix / W2, ox = pix - oy * W2;
    const float sy = fmaxf(((float)oy + 0.5f) * 0.5f - 0.5f, 0.0f), sx = fmaxf(((float)ox + 0.5f) * 0.5f - 0.5f, 0.0f);
    const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const f16x8 v00 = *reinterpret_cast<const f16x8*>(x + ((size_t)y0 * w + x0) * c + g * 8);
    const f16x8 v01 = *reinterpret_cast<const f16x8*>(x + ((size_t)y0 * w + x1) * c + g * 8);
    const f16x8 v10 = *reinterpret_cast<const f16x8*>(x + ((size_t)y1 * w + x0) * c + g * 8);
    const f16x8 v11 = *reinterpret_cast<const f16x8*>(x + ((size_t)y1 * w + x1) * c + g * 8);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float top = (1.0f - lx) * (float)v00[j] + lx * (float)v01[j];
        const float bot = (1.0f - lx) * (float)v10[j] + lx * (float)v11[j];
        o[j] = (f16)(((1.0f - ly) * top + ly * bot) * mul);
    }
    *reinterpret_cast<f16x8*>(out + (size_t)pix * c + g * 8) = o;
}

extern "C" int refvsr_up2_bilinear_nhwc16(const void* x, int h, int w, int c, float mul, void* out, void* stream) {
    RV_CHECK(x && out && h > 0 && w > 0 && c % 8 == 0, "up2_bilinear: bad args");
    hipLaunchKernelGGL(up2_bilinear_kernel, dim3(rv_cdiv(4 * h * w * (c / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)x, h, w, c, mul, (f16*)out);
    RV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// out = feat * sigmoid(attn) * 2 + attn_add   (TSAFusion output, edvr_net.py:294-299)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void tsa_blend_kernel(const f16* __restrict__ feat, const f16* __restrict__ attn, const f16* __restrict__ add,
                                 f16* __restrict__ out, size_t n8) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const f16x8 f = reinterpret_cast<const f16x8*>(feat)[i], a = reinterpret_cast<const f16x8*>(attn)[i],
                d = reinterpret_cast<const f16x8*>(add)[i];
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)((float)f[j] * (1.0f / (1.0f + __expf(-(float)a[j]))) * 2.0f + (float)d[j]);
    reinterpret_cast<f16x8*>(out)[i] = o;
}

extern "C" int refvsr_tsa_blend(const void* feat, const void* attn, const void* add, size_t n, void* out, void* stream) {
    RV_CHECK(feat && attn && add && out && n % 8 == 0 && n > 0, "tsa_blend: bad args");
    const size_t n8 = n / 8;
    hipLaunchKernelGGL(tsa_blend_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)feat, (const f16*)attn, (const f16*)add, (f16*)out, n8);
    RV_LAUNCH_CHECK();
    return 0;
}
