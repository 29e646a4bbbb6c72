// fp32 direct convolution on planar maps (VALU, exact fp32 FMA).  Used where the result feeds the
// discontinuous arg-max of the reference matching (VGG19 conv1_1/conv1_2[/conv2_1] + 1x1 map,
// attention.py:28-42; MeanShift 1x1, common.py:84-94) and for the 2->16 confidence convs whose
// inputs are fp32 confidence maps (RefVSR.py:47-52).
//
// One workgroup = 16x16 output pixels x CO_T output channels.  Input channels are walked in groups
// of CI_T: the (16*stride + k - 1)^2 x CI_T input patch and the matching weight slab are staged in
// LDS; weights are read back as 16-byte LDS broadcasts (4 output channels per ds_read_b128).
#include "common.h"

#define CD_T 16
#define CD_CO 16
#define CD_CI 8

struct DirectArgs {
    const float* src; const float* wgt; const float* bias; void* out;
    int cin, h, w, cout, ks, stride, pad, ho, wo;
    float slope;
    int out_nhwc16, out_c;
};

__global__ __launch_bounds__(256) void conv_direct_kernel(DirectArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int ox = blockIdx.x * CD_T + tx, oy = blockIdx.y * CD_T + ty;
    const int co0 = blockIdx.z * CD_CO;
    const int PH = (CD_T - 1) * a.stride + a.ks, PW = PH;
    const int iy0 = blockIdx.y * CD_T * a.stride - a.pad;
    const int ix0 = blockIdx.x * CD_T * a.stride - a.pad;
    const size_t plane = (size_t)a.h * a.w;
    const int kk = a.ks * a.ks;
    float* patch = smem;                               // [CD_CI][PH][PW]
    float* wl = smem + CD_CI * PH * PW;                // [CD_CI][kk][CD_CO]  (co fastest: one 64-byte broadcast row per tap)

    float acc[CD_CO];
#pragma unroll
    for (int c = 0; c < CD_CO; ++c) acc[c] = 0.0f;

    for (int ci0 = 0; ci0 < a.cin; ci0 += CD_CI) {
        const int nci = min(CD_CI, a.cin - ci0);
        __syncthreads();
        for (int i = threadIdx.x; i < nci * PH * PW; i += 256) {
            const int ci = i / (PH * PW);
            const int r = (i - ci * PH * PW) / PW;
            const int c = i - ci * PH * PW - r * PW;
            const int iy = iy0 + r, ix = ix0 + c;
            float v = 0.0f;
            if (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) v = a.src[(ci0 + ci) * plane + (size_t)iy * a.w + ix];
            patch[i] = v;
        }
        for (int i = threadIdx.x; i < nci * kk * CD_CO; i += 256) {
            const int c = i % CD_CO;
            const int t = (i / CD_CO) % kk;
            const int ci = i / (CD_CO * kk);
            wl[i] = (co0 + c < a.cout) ? a.wgt[((size_t)(co0 + c) * a.cin + (ci0 + ci)) * kk + t] : 0.0f;
        }
        __syncthreads();
        for (int ci = 0; ci < nci; ++ci) {
            const float* pp = patch + ci * PH * PW + (ty * a.stride) * PW + tx * a.stride;
            const float* wc = wl + ci * kk * CD_CO;
            for (int ky = 0; ky < a.ks; ++ky) {
                for (int kx = 0; kx < a.ks; ++kx) {
                    const float xv = pp[ky * PW + kx];
                    const f32x4* w4 = reinterpret_cast<const f32x4*>(wc + (ky * a.ks + kx) * CD_CO);
#pragma unroll
                    for (int c4 = 0; c4 < CD_CO / 4; ++c4) {
                        const f32x4 wv = w4[c4];            // same address in every lane: LDS broadcast
                        acc[c4 * 4 + 0] = fmaf(xv, wv[0], acc[c4 * 4 + 0]);
                        acc[c4 * 4 + 1] = fmaf(xv, wv[1], acc[c4 * 4 + 1]);
                        acc[c4 * 4 + 2] = fmaf(xv, wv[2], acc[c4 * 4 + 2]);
                        acc[c4 * 4 + 3] = fmaf(xv, wv[3], acc[c4 * 4 + 3]);
                    }
                }
            }
        }
    }
    if (ox >= a.wo || oy >= a.ho) return;
    const size_t opix = (size_t)oy * a.wo + ox;
    if (a.out_nhwc16) {
        f16* o = reinterpret_cast<f16*>(a.out) + opix * a.out_c + co0;
        if (co0 + CD_CO <= a.cout) {
            f16x8 v0, v1;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v0[c] = (f16)rv_lrelu(acc[c] + a.bias[co0 + c], a.slope);
                v1[c] = (f16)rv_lrelu(acc[8 + c] + a.bias[co0 + 8 + c], a.slope);
            }
            *reinterpret_cast<f16x8*>(o) = v0;
            *reinterpret_cast<f16x8*>(o + 8) = v1;
        } else {
#pragma unroll
            for (int c = 0; c < CD_CO; ++c)
                if (co0 + c < a.cout) o[c] = (f16)rv_lrelu(acc[c] + a.bias[co0 + c], a.slope);
        }
    } else {
        float* o = reinterpret_cast<float*>(a.out);
#pragma unroll
        for (int c = 0; c < CD_CO; ++c)
            if (co0 + c < a.cout)
                o[(size_t)(co0 + c) * a.ho * a.wo + opix] = rv_lrelu(acc[c] + a.bias[co0 + c], a.slope);
    }
}

extern "C" int refvsr_conv_direct_f32(const float* src, int cin, int h, int w,
                                      const float* wgt, const float* bias, int cout, int ksize, int stride, int pad,
                                      float act_slope, void* out, int out_nhwc16, int out_c, void* stream) {
    RV_CHECK(src && wgt && bias && out && cin > 0 && cout > 0 && h > 0 && w > 0, "conv_direct: bad args");
    RV_CHECK(ksize >= 1 && ksize <= 7 && stride >= 1 && stride <= 2 && pad >= 0, "conv_direct: bad geometry");
    RV_CHECK(!out_nhwc16 || out_c >= cout, "conv_direct: out_c < cout");
    DirectArgs a;
    a.src = src; a.wgt = wgt; a.bias = bias; a.out = out;
    a.cin = cin; a.h = h; a.w = w; a.cout = cout; a.ks = ksize; a.stride = stride; a.pad = pad;
    a.ho = (h + 2 * pad - ksize) / stride + 1;
    a.wo = (w + 2 * pad - ksize) / stride + 1;
    a.slope = act_slope; a.out_nhwc16 = out_nhwc16; a.out_c = out_c;
    const int PH = (CD_T - 1) * stride + ksize;
    const size_t lds = ((size_t)CD_CI * PH * PH + (size_t)CD_CI * ksize * ksize * CD_CO) * sizeof(float);
    RV_CHECK(!out_nhwc16 || out_c % 8 == 0, "conv_direct: nhwc16 output needs out_c %% 8 == 0");
    dim3 grid(rv_cdiv(a.wo, CD_T), rv_cdiv(a.ho, CD_T), rv_cdiv(cout, CD_CO));
    hipLaunchKernelGGL(conv_direct_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
    RV_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// 1x1 conv, C_in -> 16, LeakyReLU, fp32 HWC in -> planar fp32 out: the `map64` / `map128` block that ends FeatureMatching's feature
// extractor (RefVSR_/attention.py:41-42: BasicBlock(default_conv, 64 | 128, 16, 1) + LeakyReLU(0.2)).  On the generic kernel's
// fp32 mode this was 118 us at 270 x 480 (0.28 TB/s: 33 MB read for 0.27 GFLOP); it is HBM work: a workgroup stages 64 pixels x
// C_in floats with coalesced 16-byte loads (row stride C_in + 1 floats: odd, so the per-pixel reads of the compute phase are
// bank-conflict free), thread (pixel, og) accumulates outputs 4 og .. 4 og + 3 over the channels in order (fp32 FMA chains, one float4
// broadcast read of the weights per channel), stores are 256-byte row segments of the planar output.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1x1_f32_kernel(const float* __restrict__ src, int cin, int npix, const float* __restrict__ wgt,
                                                         const float* __restrict__ bias, float slope, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float c1_lds[];
    const int ST = cin + 1;                                          // floats per staged pixel (odd: lanes = pixels hit distinct banks)
    float* xs = c1_lds;                                              // [64][ST]
    float* ws = c1_lds + ((64 * ST + 3) & ~3);                       // [cin][16]: output fastest, 16-byte aligned
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * 64;
    const int q4 = cin >> 2;                                         // float4 per pixel
    for (int i = tid; i < 64 * q4; i += 256) {
        const int px = i / q4, c4 = i - px * q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + px < npix) v = *reinterpret_cast<const float4*>(src + (size_t)(p0 + px) * cin + c4 * 4);
        float* d = xs + px * ST + c4 * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int i = tid; i < cin * 16; i += 256) {
        const int c = i >> 4, o = i & 15;
        ws[i] = wgt[o * cin + c];
    }
    __syncthreads();
    const int px = tid & 63, og = tid >> 6;
    const float* xp = xs + px * ST;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int c = 0; c < cin; ++c) {
        const float x = xp[c];
        const float4 w4 = *reinterpret_cast<const float4*>(ws + c * 16 + og * 4);
        a0 = fmaf(x, w4.x, a0); a1 = fmaf(x, w4.y, a1); a2 = fmaf(x, w4.z, a2); a3 = fmaf(x, w4.w, a3);
    }
    if (p0 + px < npix) {
        float* o = out + (size_t)(og * 4) * npix + p0 + px;
        o[0] = rv_lrelu(a0 + bias[og * 4 + 0], slope);
        o[(size_t)npix] = rv_lrelu(a1 + bias[og * 4 + 1], slope);
        o[2 * (size_t)npix] = rv_lrelu(a2 + bias[og * 4 + 2], slope);
        o[3 * (size_t)npix] = rv_lrelu(a3 + bias[og * 4 + 3], slope);
    }
}

extern "C" int refvsr_conv1x1_f32(const float* src, int cin, int h, int w, const float* wgt, const float* bias, float act_slope,
                                  float* out, void* stream) {
    RV_CHECK(src && wgt && bias && out && h > 0 && w > 0, "conv1x1_f32: bad args");
    RV_CHECK(cin >= 4 && cin <= 256 && cin % 4 == 0, "conv1x1_f32: cin must be a multiple of 4 in [4, 256] (got %d)", cin);
    RV_CHECK(((uintptr_t)src & 15) == 0, "conv1x1_f32: src must be 16-byte aligned");
    RV_CHECK(act_slope >= 0.f && act_slope <= 1.f, "conv1x1_f32: activation slope must lie in [0, 1]");
    RV_CHECK(refvsr_init() == 0, "init failed");
    const int npix = h * w;
    const size_t lds = ((size_t)((64 * (cin + 1) + 3) & ~3) + (size_t)cin * 16) * sizeof(float);
    static bool attr_done[RV_MAX_DEVICES] = {};
    const int dev = rv_device();
    if (!attr_done[dev]) {
        RV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL(conv1x1_f32_kernel, dim3(rv_cdiv(npix, 64)), dim3(256), lds, (hipStream_t)stream, src, cin, npix, wgt, bias,
                       act_slope, out);
    RV_LAUNCH_CHECK();
    return 0;
}
