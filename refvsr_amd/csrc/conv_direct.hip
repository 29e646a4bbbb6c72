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
