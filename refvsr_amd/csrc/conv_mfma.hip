// Implicit-GEMM convolution on CDNA4 matrix cores (v_mfma_f32_16x16x32_f16), nhwc16 activations.
//
//   D[cout][pixel] = sum_k  W[cout][k] * X[k][pixel],   k = (tap, channel)   (fp32 accumulate)
//
// * A operand = packed weights (rows = 16 output channels), B operand = 16 output pixels of one image row.  With
//   this orientation a lane ends up holding 4 consecutive output channels of ONE pixel, so the epilogue stores
//   8-byte channel vectors straight into the HWC map.
// * One workgroup = 4 waves = an (2*TILES) x 32 output-pixel tile.  The input tile (+halo, zero padded at the frame
//   border, two concatenated sources) is staged through LDS with coalesced 16-byte loads of HWC rows; every tap of
//   every output pixel is then an LDS read.  LDS pixel stride is odd (in 16-byte slots); together with the K-block
//   order and the pixel permutation of common.h (rv_kslot, rv_pix16) the ds_read_b128 of a B fragment is bank
//   conflict free.
// * K is walked in 32-deep steps; per step lane l consumes K-slot 4*step + (l>>4), an 8-channel group of one tap.
//   slot -> LDS byte offset comes from a small table built per workgroup, so kernel size, stride and channel count
//   are runtime values (one kernel serves 1x1..7x7).  The K loop is software pipelined (two fragment sets).
// * Weights are pre-packed on the host in exact fragment order.  fp16 mode: every weight is carried as hi + lo halves
//   ([kstep][mtile][hi|lo][lane][8 halfs], two MFMAs per B fragment): a plain fp16 weight rounding is a *systematic*
//   perturbation of the network and was measured to dominate the PSNR-parity error (tools/precision_sim.py); hi+lo
//   gives ~22-bit weights for one extra MFMA.
// * RESIDENT mode (template flag; up to CONV_RES_MAX K-steps, i.e. every 3x3 conv with <= 56 input channels, the 5x5
//   3->32 and the first SPyNet 7x7): the whole weight set stays in LDS and the workgroup is persistent -- it walks
//   pixel tiles (XCD-banded order) and fetches the NEXT tile into registers while the matrix pipe works on the current
//   one.  On the 2x / HR maps the per-CU vector-memory path bounds these convs: re-fetching 29-57 KB of weights per
//   8x32-pixel tile was more than half of its traffic, and a fresh workgroup per tile exposed the staging latency.
//   Otherwise (SPyNet 7x7 32/64-channel layers, 5x5 64-channel predictors, VGG 64->64) weights stream through LDS in
//   chunks of CONV_CH K-steps, register-prefetched, one workgroup per tile.
// * F32 mode (template flag): the same walk on v_mfma_f32_16x16x4_f32 with fp32 HWC activations and fp32 weights
//   ([kstep][mtile][lane][4 floats]; a K-block is 4 channels, 4 MFMAs per 16-byte fragment) -- bitwise an fp32 FMA
//   chain.  Used for the VGG feature extractor of the matching, whose arg-max is discontinuous in the features.
// * Epilogue fuses bias, (leaky)ReLU, alpha-multiply, residual, post-activation, pixel-shuffle or a planar fp32
//   store with residual / constant / clamp.
// * Gather mode (template flag, chosen when the staged tile would not fit LDS, i.e. the stride-4/8 5x5 offset
//   predictors of the HD configs, alignment.py:20 with stride = ks): windows of neighbouring output pixels do not
//   overlap there, so B fragments are read straight from global memory (16 bytes per lane, zero for out-of-frame
//   taps) and only the weights go through LDS.
//
// Replaces nn.Conv2d call sites listed in include/refvsr_hip.h.
#include <type_traits>

#include "common.h"
#include <stdlib.h>

#define CONV_CH 8          // K-steps of weights per streamed LDS chunk (chunked convs)
#define CONV_RES_MAX 16    // K-steps a resident weight set may have (persistent convs)
#define CONV_XPF 8         // uint4 prefetch registers per thread for the next input tile (persistent convs)
#define CONV_TW 32         // output tile width in pixels

struct ConvArgs {
    const unsigned char* src0; const unsigned char* src1;   // HWC maps: f16 (8 ch / 16 B) or f32 (4 ch / 16 B)
    int c0, c1, ncg0, ncg, ps;       // channel counts; ncg* = 16-byte groups per pixel; ps = LDS pixel stride in slots (odd)
    int pixb0, pixb1;                // bytes per pixel of src0 / src1
    int h_in, w_in, h_out, w_out;
    int ks, stride, pad;
    int LH, LW;                      // LDS input tile extent in pixels
    int G, S;                        // valid K-blocks, K-steps
    float inv_ncg;
    const uint4* wpack; const float* bias;
    int cout;
    float act_slope, post_slope;
    const unsigned char* mul; int mul_c;
    const unsigned char* res; int res_c;
    int out_mode; void* out; int out_c;
    const float* res_planar; float add_const, clamp_lo, clamp_hi;
    int tab_bytes, wl_bytes;         // LDS carve sizes
    int gather;                      // 1: no LDS input tile, B fragments gathered from global memory
    int tiles_x, n_xy;               // pixel tiles per row / per frame
    int prefetch;                    // resident kernels: 1 = next tile prefetched into registers during the K loop
    int grid;                        // gridDim.x (resident kernels: rv_tile_range)
    int ring;                        // streamed kernels: LDS ring slots of CONV_CH K-steps each (2..4)
    int batch;                       // images per launch (blockIdx.y); byte strides between the images of each map
    unsigned long long bs_src0, bs_src1, bs_out, bs_res_planar;
};

// EPI = 0: general epilogue (every output mode, fp16 / fp32 maps).  EPI = 1: the fp16 HWC store with optional alpha
// multiply / residual and activations given as slopes in [0, 1] -- the path of ~90 % of the launches -- written for a low
// instruction count: the PMC split of round 1 (33 % of wave cycles issuing at 3 waves per SIMD = a saturated issue
// port, MFMA 19-30 % busy) says these kernels are bound by the NUMBER of instructions around the K loop, not by memory
// or the matrix pipe.
// NW: waves per workgroup.  4 by default.  Where LDS admits ONE workgroup per CU -- the C = 48 weight sets of
// RefVSR_MFID(_8K): 84 KB resident, or a 65 KB two-source tile next to the streamed chunk -- a 4-wave workgroup leaves a
// SIMD with a single wave and nothing to interleave: those launches use 16 waves on a 16 x 32 tile (resident, where it fits)
// or 8 waves on the same 8 x 32 tile (two pixel groups per wave).
// (Rounds 3-4 carried a WARP variant here -- the inter-frame warp evaluated while the tile is staged, bit-identical to warp + conv:
// measured slower in every configuration, 169 vs 176 frames/s, profiles/r03_fused_warp_ab.txt -- removed in round 5.)
template <int MT, int TILES, bool F32, bool GATHER, bool RESIDENT, int EPI = 0, int NW = 4, bool HI1 = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 4))) void conv_mfma_kernel(ConvArgs p) {
    constexpr int NT = NW * 64;                      // threads per workgroup
    static_assert(!(GATHER && RESIDENT), "gather mode streams its weights");
    static_assert(EPI == 0 || (RESIDENT && !F32), "the lean epilogue is built for the resident fp16 kernels");
    static_assert(!HI1 || (!F32 && !RESIDENT && !GATHER), "single-fp16 weights are built for the streamed fp16 kernels");
    constexpr int WFR = (F32 || HI1) ? 1 : 2;        // 1 KiB weight fragments per (kstep, mtile): fp32 | fp16 hi+lo | fp16 hi only (HI1)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* tab = reinterpret_cast<int*>(smem);
    unsigned char* wl = smem + p.tab_bytes;
    unsigned char* tile = wl + p.wl_bytes;           // [table][weights: resident set | one chunk][input tile]

    // the kernel arguments, fetched by ONE batch of scalar loads (hipcc otherwise loads the fields of the by-value struct where
    // they are first used: a chain of dependent s_load / s_waitcnt round trips at the head of every launch)
    asm volatile("" :: "s"(p.src0), "s"(p.src1), "s"(p.wpack), "s"(p.bias), "s"(p.out), "s"(p.mul), "s"(p.res), "s"(p.c0),
                 "s"(p.ncg0), "s"(p.ncg), "s"(p.ps), "s"(p.pixb0), "s"(p.pixb1), "s"(p.h_in), "s"(p.w_in), "s"(p.h_out),
                 "s"(p.w_out), "s"(p.ks), "s"(p.stride), "s"(p.pad), "s"(p.LH), "s"(p.LW), "s"(p.G), "s"(p.S), "s"(p.inv_ncg),
                 "s"(p.cout), "s"(p.tab_bytes), "s"(p.wl_bytes), "s"(p.tiles_x), "s"(p.n_xy), "s"(p.grid));
    if (p.batch > 1) {                               // image blockIdx.y of the batch (uniform: scalar pointer arithmetic)
        const unsigned long long bi = blockIdx.y;
        p.src0 += bi * p.bs_src0;
        if (p.src1) p.src1 += bi * p.bs_src1;
        p.out = reinterpret_cast<unsigned char*>(p.out) + bi * p.bs_out;
        if (p.res_planar) p.res_planar = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p.res_planar) + bi * p.bs_res_planar);
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = lane >> 4;
    const int lr = lane & 15;
    const int lp = (p.stride == 1) ? rv_pix16(lr) : lr;      // pixel of the 16-pixel tile this lane's MFMA column holds
    constexpr int TH = NW * TILES / 2;               // output rows of the workgroup's tile (two 16-pixel groups per row)
    const int zg = blockIdx.z;

    // ---- K-slot -> LDS byte offset table (K order: common.h:rv_kslot) ---------------------------
    for (int g = tid; g < p.S * 4; g += NT) {
        int off, slot;
        if (g < p.G) {
            const int tap = (int)(((float)g + 0.5f) * p.inv_ncg);
            const int cg = g - tap * p.ncg;
            const int ty = tap / p.ks;
            const int tx = tap - ty * p.ks;
            off = GATHER ? (ty | (tx << 8) | (cg << 16)) : ((ty * p.LW + tx) * p.ps + cg) * 16;
            slot = rv_kslot(ty, tx, cg, p.ks, p.ncg);
        } else {                                   // zero-weight blocks: any readable offset of the partner's parity
            const int j = g - p.G;
            slot = rv_kpad_slot(j, p.ks, p.ncg);
            off = GATHER ? -1 : (slot < p.G ? 0 : (p.ncg > 1 ? 16 : (p.ks > 1 ? p.ps * 16 : 0)));
        }
        tab[slot] = off;
    }

    // ---- per-lane base offsets of this wave's pixel tiles (tile mode: independent of the tile origin) -------------
    int pbase[TILES];        // tile mode: LDS byte offset of the pixel's window origin; gather mode: packed (iy0, ix0)
    if constexpr (!GATHER) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int ti = wave * TILES + t;
            pbase[t] = (((ti >> 1) * p.stride) * p.LW + ((ti & 1) * 16 + lp) * p.stride) * p.ps * 16;
        }
    }
    auto load_b = [&](int pb, int toff) -> uint4 {
        if constexpr (!GATHER) return *reinterpret_cast<const uint4*>(tile + pb + toff);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (GATHER && toff >= 0) {
            const int iy = (pb >> 16) - 4096 + (toff & 0xff);
            const int ix = (pb & 0xffff) - 4096 + ((toff >> 8) & 0xff);
            const int cg = toff >> 16;
            if (iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in) {
                const size_t pix = (size_t)iy * p.w_in + ix;
                v = (cg < p.ncg0) ? *reinterpret_cast<const uint4*>(p.src0 + pix * p.pixb0 + cg * 16)
                                  : *reinterpret_cast<const uint4*>(p.src1 + pix * p.pixb1 + (cg - p.ncg0) * 16);
            }
        }
        return v;
    };

    const uint4* wsrc = p.wpack + (size_t)zg * p.S * MT * WFR * 64;

    // bias of this lane's output channels: fetched once, ahead of the K loop
    float4 bias_r[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co0 = (zg * MT + m) * 16 + q * 4;
        bias_r[m] = *reinterpret_cast<const float4*>(p.bias + co0);   // the bias array is padded to nz*MT*16 floats
    }

    // ---- K loop, software pipelined: the fragments of step s+1 are read (LDS, or global memory in gather mode)
    // while the MFMAs of step s are in the matrix pipe --------------------------------------------------------------
    constexpr int NA = MT * WFR;                                 // 16-byte A fragments per K-step
    f32x4 acc[MT][TILES];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < TILES; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto load_frag = [&](const unsigned char* wcur, const int sl, const int toff, uint4 (&a)[NA], uint4 (&b)[TILES]) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            a[i] = *reinterpret_cast<const uint4*>(wcur + ((size_t)(sl * NA + i) * 64 + lane) * 16);
#pragma unroll
        for (int t = 0; t < TILES; ++t) b[t] = load_b(pbase[t], toff);
    };
    auto mfma_step = [&](const uint4 (&a)[NA], const uint4 (&b)[TILES]) {
        if constexpr (F32) {
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&b[t]);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(&a[m]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[m][t], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < WFR; ++h)                          // all hi products, then all lo: no back-to-back
#pragma unroll
                for (int t = 0; t < TILES; ++t) {                  // MFMAs on one accumulator
                    const f16x8 bv = *reinterpret_cast<const f16x8*>(&b[t]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const f16x8 av = *reinterpret_cast<const f16x8*>(&a[m * WFR + h]);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[m][t], 0, 0, 0);
                    }
                }
        }
    };
    // K-steps s0 .. s0+ns-1 with their weight fragments at wcur (fragment index relative to s0)
    auto compute_steps = [&](const unsigned char* wcur, const int s0, const int ns) {
        const int* tq = tab + s0 * 4 + q;
        const int last = ns - 1;
        // two fragment sets, the loop unrolled by two: set 1 is read while set 0 is in the matrix pipe and vice versa;
        // the table entry of a step is fetched half an iteration before its fragments.  Steps past the end re-read the
        // last one (dropped), so every register has one unconditional definition per iteration.
        uint4 a0[NA], b0[TILES], a1[NA], b1[TILES];
        load_frag(wcur, 0, tq[0], a0, b0);
        int t1 = tq[min(1, last) * 4];
        for (int sl = 0; sl < ns; sl += 2) {
            const int s1 = min(sl + 1, last), s2 = min(sl + 2, last), s3 = min(sl + 3, last);
            const int t2 = tq[s2 * 4];
            load_frag(wcur, s1, t1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);                     // reads stay above the MFMAs they overlap with
            mfma_step(a0, b0);
            const int t3 = tq[s3 * 4];
            load_frag(wcur, s2, t2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (sl + 1 < ns) mfma_step(a1, b1);
            t1 = t3;
        }
    };

    // ---- epilogue of one tile; `tide` = thread id, opaque per tile in the persistent walk (otherwise hipcc hoists the
    // whole epilogue address math -- dozens of 64-bit products -- out of the tile loop and keeps it in registers) -----
    auto epilogue = [&](const int ty0, const int tx0, const int tide) {
    const int lane_e = tide & 63, wave_e = tide >> 6, q_e = lane_e >> 4;
    const int lp_e = (p.stride == 1) ? rv_pix16(lane_e & 15) : (lane_e & 15);
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const int ti = wave_e * TILES + t;
        const int oy = ty0 + (ti >> 1);
        const int ox = tx0 + (ti & 1) * 16 + lp_e;
        if (oy >= p.h_out || ox >= p.w_out) continue;
        const size_t opix = (size_t)oy * p.w_out + ox;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co0 = (zg * MT + m) * 16 + q_e * 4;
            if (co0 >= p.cout) {
                // channel padding of an HWC map whose channel count is not a multiple of 8 (C = 36 of RefVSR_IR -> stride
                // 40): consumers multiply the padding by zero weights, so it must hold zeros, not allocator garbage
                if (p.out_mode == REFVSR_OUT_NHWC16 && co0 < p.out_c) {
                    if constexpr (F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + opix * p.out_c + co0) = (f32x4){0.f, 0.f, 0.f, 0.f};
                    else *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.out) + opix * p.out_c + co0) = (f16x4){0, 0, 0, 0};
                }
                continue;
            }
            float y[4];
            const float4 bv = bias_r[m];
            y[0] = acc[m][t][0] + bv.x; y[1] = acc[m][t][1] + bv.y;
            y[2] = acc[m][t][2] + bv.z; y[3] = acc[m][t][3] + bv.w;
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = rv_lrelu(y[i], p.act_slope);
            if (p.mul) {
                float mv[4];
                if constexpr (F32) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(p.mul + (opix * p.mul_c + co0) * 4);
                    mv[0] = t4[0]; mv[1] = t4[1]; mv[2] = t4[2]; mv[3] = t4[3];
                } else {
                    const f16x4 t4 = *reinterpret_cast<const f16x4*>(p.mul + (opix * p.mul_c + co0) * 2);
                    mv[0] = (float)t4[0]; mv[1] = (float)t4[1]; mv[2] = (float)t4[2]; mv[3] = (float)t4[3];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] *= mv[i];
            }
            if (p.res) {
                float rv[4];
                if constexpr (F32) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(p.res + (opix * p.res_c + co0) * 4);
                    rv[0] = t4[0]; rv[1] = t4[1]; rv[2] = t4[2]; rv[3] = t4[3];
                } else {
                    const f16x4 t4 = *reinterpret_cast<const f16x4*>(p.res + (opix * p.res_c + co0) * 2);
                    rv[0] = (float)t4[0]; rv[1] = (float)t4[1]; rv[2] = (float)t4[2]; rv[3] = (float)t4[3];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] += rv[i];
            }
            if (p.post_slope != 1.0f) {
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = rv_lrelu(y[i], p.post_slope);
            }
            if (p.out_mode == REFVSR_OUT_NHWC16) {
                if constexpr (F32) {
                    f32x4 o = {y[0], y[1], y[2], y[3]};
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + opix * p.out_c + co0) = o;
                } else {
                    f16x4 o = {(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
                    *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.out) + opix * p.out_c + co0) = o;
                }
            } else if (p.out_mode == REFVSR_OUT_NHWC16_SHUFFLE2) {
                // packed row r = sub*C + c  <->  conv channel c*4 + sub, sub = dy*2 + dx
                const int C = p.cout >> 2;
                const int sub = co0 / C;
                const int c = co0 - sub * C;
                const size_t op = (size_t)(2 * oy + (sub >> 1)) * (2 * p.w_out) + (2 * ox + (sub & 1));
                f16x4 o = {(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
                *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.out) + op * p.out_c + c) = o;
                if (c + 4 == C && p.out_c > C)             // zero the channel padding of the shuffled map (see above)
                    *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.out) + op * p.out_c + C) = (f16x4){0, 0, 0, 0};
            } else {
                const size_t plane = (size_t)p.h_out * p.w_out;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = co0 + i;
                    if (co < p.cout) {
                        float v = y[i];
                        if (p.res_planar) v += p.res_planar[co * plane + opix];
                        v += p.add_const;
                        if (p.clamp_lo < p.clamp_hi) v = fminf(fmaxf(v, p.clamp_lo), p.clamp_hi);
                        reinterpret_cast<float*>(p.out)[co * plane + opix] = v;
                    }
                }
            }
        }
    }
    };

    // ---- EPI = 1: bias, activation as max(y, slope*y), optional alpha multiply / residual, fp16 HWC store; 32-bit element
    // offsets (the host checks the map sizes) ---------------------------------------------------------------------------
    auto epilogue_lean = [&](const int ty0, const int tx0, const int tide) {
        const int lane_e = tide & 63, wave_e = tide >> 6, q_e = lane_e >> 4;
        const int lp_e = (p.stride == 1) ? rv_pix16(lane_e & 15) : (lane_e & 15);
        const f16* mulp = reinterpret_cast<const f16*>(p.mul);
        const f16* resp = reinterpret_cast<const f16*>(p.res);
        f16* outp = reinterpret_cast<f16*>(p.out);
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int ti = wave_e * TILES + t;
            const int oy = ty0 + (ti >> 1);
            const int ox = tx0 + (ti & 1) * 16 + lp_e;
            if (oy >= p.h_out || ox >= p.w_out) continue;
            const unsigned opix = (unsigned)(oy * p.w_out + ox);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = (zg * MT + m) * 16 + q_e * 4;
                if (co0 >= p.cout) {
                    if (co0 < p.out_c) *reinterpret_cast<f16x4*>(outp + (opix * (unsigned)p.out_c + (unsigned)co0)) = (f16x4){0, 0, 0, 0};
                    continue;
                }
                const float4 bv = bias_r[m];
                float y0 = acc[m][t][0] + bv.x, y1 = acc[m][t][1] + bv.y, y2 = acc[m][t][2] + bv.z, y3 = acc[m][t][3] + bv.w;
                y0 = fmaxf(y0, y0 * p.act_slope); y1 = fmaxf(y1, y1 * p.act_slope);
                y2 = fmaxf(y2, y2 * p.act_slope); y3 = fmaxf(y3, y3 * p.act_slope);
                if (mulp) {
                    const f16x4 t4 = *reinterpret_cast<const f16x4*>(mulp + (opix * (unsigned)p.mul_c + (unsigned)co0));
                    y0 *= (float)t4[0]; y1 *= (float)t4[1]; y2 *= (float)t4[2]; y3 *= (float)t4[3];
                }
                if (resp) {
                    const f16x4 t4 = *reinterpret_cast<const f16x4*>(resp + (opix * (unsigned)p.res_c + (unsigned)co0));
                    y0 += (float)t4[0]; y1 += (float)t4[1]; y2 += (float)t4[2]; y3 += (float)t4[3];
                }
                if (p.post_slope != 1.0f) {
                    y0 = fmaxf(y0, y0 * p.post_slope); y1 = fmaxf(y1, y1 * p.post_slope);
                    y2 = fmaxf(y2, y2 * p.post_slope); y3 = fmaxf(y3, y3 * p.post_slope);
                }
                const f16x4 o = {(f16)y0, (f16)y1, (f16)y2, (f16)y3};
                *reinterpret_cast<f16x4*>(outp + (opix * (unsigned)p.out_c + (unsigned)co0)) = o;
            }
        }
    };

    if constexpr (RESIDENT) {
        // ================= persistent workgroup: resident weights, register-prefetched tiles =====================
        // weights -> LDS in batches of eight 16-byte loads per thread: every load of a batch is in flight before the first
        // LDS store (a plain copy loop costs one memory round trip per 4 KB)
        {
            const int n16 = p.S * MT * WFR * 64;
            uint4* d = reinterpret_cast<uint4*>(wl);
            for (int i0 = 0; i0 < n16; i0 += 8 * NT) {
                uint4 w0, w1, w2, w3, w4, w5, w6, w7;
                const int last = n16 - 1, i = i0 + tid;
                w0 = wsrc[min(i, last)];              w1 = wsrc[min(i + NT, last)];
                w2 = wsrc[min(i + 2 * NT, last)];     w3 = wsrc[min(i + 3 * NT, last)];
                w4 = wsrc[min(i + 4 * NT, last)];     w5 = wsrc[min(i + 5 * NT, last)];
                w6 = wsrc[min(i + 6 * NT, last)];     w7 = wsrc[min(i + 7 * NT, last)];
                asm volatile("" ::: "memory");
                if (i < n16) d[i] = w0;
                if (i + NT < n16) d[i + NT] = w1;
                if (i + 2 * NT < n16) d[i + 2 * NT] = w2;
                if (i + 3 * NT < n16) d[i + 3 * NT] = w3;
                if (i + 4 * NT < n16) d[i + 4 * NT] = w4;
                if (i + 5 * NT < n16) d[i + 5 * NT] = w5;
                if (i + 6 * NT < n16) d[i + 6 * NT] = w6;
                if (i + 7 * NT < n16) d[i + 7 * NT] = w7;
            }
        }
        int tl, k_hi;                                              // this workgroup's tiles (common.h:rv_tile_range)
        rv_tile_range(p.n_xy, p.grid, tl, k_hi);
        constexpr int k_step = 1;
        // input-tile chunk k of this thread: 16 bytes = (row r, column c, channel group cg) of the LH x LW tile.  Everything
        // that does not depend on the tile origin is computed ONCE per thread: xq = r | c << 5 | (cg within its source) << 12 |
        // src1 << 19 | cg << 20 (-1 past the tile).  Per tile a chunk then costs ~20 instructions (bounds, offset from the tile's
        // origin pixel, base select, 64-bit add, load) instead of ~58.
        const int row_chunks = p.LW * p.ncg;
        const int total = p.LH * row_chunks;                       // <= CONV_XPF * 256 (host)
        constexpr int XPF = NW == 4 ? CONV_XPF : 4;                // chunk slots per thread (capacity 2048 | 2048 | 4096)
        int xq[XPF];
        {
            const float inv_rc = 1.0f / (float)row_chunks;
#pragma unroll
            for (int k = 0; k < XPF; ++k) {
                const int idx = tid + k * NT;
                const int r = (int)(((float)idx + 0.5f) * inv_rc);
                const int i = idx - r * row_chunks;
                const int c = (int)(((float)i + 0.5f) * p.inv_ncg);
                const int cg = i - c * p.ncg;
                const bool s1 = cg >= p.ncg0;
                xq[k] = idx < total ? (r | (c << 5) | ((s1 ? cg - p.ncg0 : cg) << 12) | (s1 ? (1 << 19) : 0) | (cg << 20)) : -1;
            }
        }
        uint4 xv[XPF];
        auto x_fetch = [&](const int t) {                          // global -> registers (zero padded at the frame border)
            const int tyi = t / p.tiles_x;
            const int iy0 = tyi * TH * p.stride - p.pad;
            const int ix0 = (t - tyi * p.tiles_x) * CONV_TW * p.stride - p.pad;
            const long long org = (long long)iy0 * p.w_in + ix0;   // origin pixel (may lie outside the frame): uniform
            const unsigned char* b0 = p.src0 + org * p.pixb0;
            const unsigned char* b1 = p.src1 + org * p.pixb1;
#pragma unroll
            for (int k = 0; k < XPF; ++k) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                int e = xq[k];
                asm volatile("" : "+v"(e));                        // decode per tile: keeps the decoded fields x 8 out of the
                const int iy = iy0 + (e & 31), ix = ix0 + ((e >> 5) & 127);   // registers that live across the K loop
                if (e >= 0 && (unsigned)iy < (unsigned)p.h_in && (unsigned)ix < (unsigned)p.w_in) {
                    const bool s1 = (e >> 19) & 1;
                    const int off = ((e & 31) * p.w_in + ((e >> 5) & 127)) * (s1 ? p.pixb1 : p.pixb0) + ((e >> 12) & 127) * 16;
                    v = *reinterpret_cast<const uint4*>((s1 ? b1 : b0) + off);
                }
                xv[k] = v;
            }
        };
        auto x_park = [&]() {                                      // registers -> LDS tile
#pragma unroll
            for (int k = 0; k < XPF; ++k) {
                int e = xq[k];
                asm volatile("" : "+v"(e));
                if (e >= 0)
                    *reinterpret_cast<uint4*>(tile + (((e & 31) * p.LW + ((e >> 5) & 127)) * p.ps + ((e >> 20) & 127)) * 16) = xv[k];
            }
        };
        // Schedule per tile i:  fetch tile i+1 into registers | K loop(i) | barrier | park tile i+1 | epilogue(i) | barrier.
        // The park sits BEFORE the output stores: vmcnt retires in order, so waiting for the prefetch behind them would
        // also wait for every store of the tile.  (p.prefetch == 0, an A/B knob: the next tile is fetched and parked in
        // one go after the K loop, i.e. with its latency exposed.)
        if (tl < k_hi) { x_fetch(tl); x_park(); }
        __syncthreads();                                           // table, weights, first tile
        for (; tl < k_hi; tl += k_step) {
            const bool has_next = tl + k_step < k_hi;
            if (has_next && p.prefetch) x_fetch(tl + k_step);      // in flight during the K loop
            asm volatile("" ::: "memory");
            zero_acc();
            compute_steps(wl, 0, p.S);
            __syncthreads();                                       // every wave is done reading the LDS tile
            if (has_next) {
                if (!p.prefetch) x_fetch(tl + k_step);
                x_park();
            }
            asm volatile("" ::: "memory");
            int tide = tid;
            asm volatile("" : "+v"(tide));
            const int tyi = tl / p.tiles_x;
            if constexpr (EPI == 1) epilogue_lean(tyi * TH, (tl - tyi * p.tiles_x) * CONV_TW, tide);
            else epilogue(tyi * TH, (tl - tyi * p.tiles_x) * CONV_TW, tide);
            __syncthreads();                                       // next tile visible
        }
    } else {
        // ================= one tile per workgroup, weights streamed in chunks =======================================
        const int tl = blockIdx.x;
        const int tyi = tl / p.tiles_x;
        const int tx0 = (tl - tyi * p.tiles_x) * CONV_TW;
        const int ty0 = tyi * TH;
        // Streamed weights (!GATHER): chunks of CONV_CH K-steps go global -> LDS with global_load_lds_dwordx4 (the LDS image of a
        // chunk is its global image) through a RING of p.ring slots: chunk c + ring - 1 is issued while chunk c feeds the MFMAs, so
        // up to ring - 1 chunks are in flight and a chunk costs max(fill, compute) instead of a memory round trip (round 2: one
        // chunk prefetched into registers, two barriers per chunk -- the 2-workgroup launches of SPyNet's coarse levels streamed
        // 392 KB through one CU in 28 us).  Every wave issues exactly PPW pieces (1 KiB each) of every chunk -- the partial last
        // chunk re-reads its last valid piece into the slot's unused tail -- so that "chunk c has landed" is
        // s_waitcnt vmcnt(pieces this wave issued after it).  Raw s_barrier + explicit waits: __syncthreads() would wait for EVERY
        // outstanding LDS-DMA (its fence cannot tell them from the stores it orders).  The first ring - 1 chunks are issued
        // before the input tile is staged.
        constexpr int PPW = GATHER ? 1 : CONV_CH * MT * WFR / NW;
        static_assert(GATHER || (PPW * NW == CONV_CH * MT * WFR && PPW >= 1 && 2 * PPW <= 60), "pieces per wave and chunk");
        constexpr int SLOT = CONV_CH * MT * WFR * 1024;
        const int NS = p.ring;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const unsigned char* gw = reinterpret_cast<const unsigned char*>(wsrc) + lane * 16;
        auto issue = [&](const int c, const int slot) {
            const int np = min(CONV_CH, p.S - c * CONV_CH) * MT * WFR;                  // valid pieces of this chunk
            const unsigned char* g = gw + (size_t)c * SLOT;
            unsigned char* l = wl + slot * SLOT;
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const int pc = wv + j * NW;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + min(pc, np - 1) * 1024),
                                                 (__attribute__((address_space(3))) void*)(l + pc * 1024), 16, 0, 0);
            }
        };
        int slot_w = 0;                                                // slot the next issued chunk goes to
        if constexpr (!GATHER) {
            const int n_chunks = (p.S + CONV_CH - 1) / CONV_CH;
            for (int c = 0; c < min(NS - 1, n_chunks); ++c) { issue(c, slot_w); slot_w = slot_w + 1 == NS ? 0 : slot_w + 1; }
        }
        if constexpr (!GATHER) {                                   // stage the input tile (zero padded)
            const int iy0 = ty0 * p.stride - p.pad;
            const int ix0 = tx0 * p.stride - p.pad;
            const int row_chunks = p.LW * p.ncg;
            const float inv_rc = 1.0f / (float)row_chunks;
            const int total = p.LH * row_chunks;
            // batches of SB chunks per thread: every load of a batch is in flight before the first LDS store (one chunk per
            // iteration = one memory round trip per chunk: the 14 x 38-pixel, 64-channel tile of SPyNet's 7x7 convs took ten of
            // them, most of the kernel's time on the coarse pyramid levels).  Loads are unconditional (clamped address), the
            // zero padding is a mask.
            constexpr int SB = 8;
            for (int idx0 = tid; idx0 < total; idx0 += SB * NT) {
                uint4 sv[SB];
                int sd[SB];
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    const int idx = min(idx0 + k * NT, total - 1);
                    const int r = (int)(((float)idx + 0.5f) * inv_rc);
                    const int i = idx - r * row_chunks;
                    const int c = (int)(((float)i + 0.5f) * p.inv_ncg);
                    const int cg = i - c * p.ncg;
                    const int iy = iy0 + r;
                    const int ix = ix0 + c;
                    const bool ok = (unsigned)iy < (unsigned)p.h_in && (unsigned)ix < (unsigned)p.w_in;
                    const size_t pix = (size_t)min(max(iy, 0), p.h_in - 1) * p.w_in + min(max(ix, 0), p.w_in - 1);
                    const unsigned char* g = (cg < p.ncg0) ? p.src0 + pix * p.pixb0 + cg * 16 : p.src1 + pix * p.pixb1 + (cg - p.ncg0) * 16;
                    uint4 v = *reinterpret_cast<const uint4*>(g);
                    const unsigned keep = ok ? 0xffffffffu : 0u;
                    v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                    sv[k] = v;
                    sd[k] = ((r * p.LW + c) * p.ps + cg) * 16;
                }
#pragma unroll
                for (int k = 0; k < SB; ++k)
                    if (idx0 + k * NT < total) *reinterpret_cast<uint4*>(tile + sd[k]) = sv[k];
            }
        } else {
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const int ti = wave * TILES + t;
                const int iy = (ty0 + (ti >> 1)) * p.stride - p.pad + 4096, ix = (tx0 + (ti & 1) * 16 + lp) * p.stride - p.pad + 4096;
                pbase[t] = (iy << 16) | ix;              // biased by 4096 so both halves stay non-negative
            }
        }
        if constexpr (!GATHER) {
            // While LDS-DMA is in flight every LDS read the COMPILER knows about gets an s_waitcnt vmcnt(0) in front of it (its
            // waitcnt pass cannot tell which DMA a ds_read may alias), which would drain the ring at the first fragment read.  The
            // reads of this loop are therefore inline asm, with the lgkmcnt waits written out: after the reads of step s + 1 are
            // issued, lgkmcnt(NA + TILES) = "everything older has returned" = the fragments of step s (LDS returns in order); the
            // empty asm ties make the MFMAs depend on the wait.
            const unsigned lds_tab = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(tab);
            const unsigned lds_wl = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)wl;
            const unsigned lds_tile = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)tile;
            unsigned pbaddr[TILES];
#pragma unroll
            for (int t = 0; t < TILES; ++t) pbaddr[t] = lds_tile + (unsigned)pbase[t];
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            auto ld128 = [](const unsigned addr) {
                u32x4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
                return make_uint4(v[0], v[1], v[2], v[3]);
            };
            auto tie = [](uint4& r) {                                 // one 128-bit operand: the four registers stay a tuple, no copies
                u32x4 t = {r.x, r.y, r.z, r.w};
                asm volatile("" : "+v"(t));
                r = make_uint4(t[0], t[1], t[2], t[3]);
            };
            auto wait_set = [&](uint4 (&a)[NA], uint4 (&b)[TILES]) {
                asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(NA + TILES) : "memory");
#pragma unroll
                for (int i = 0; i < NA; ++i) tie(a[i]);
#pragma unroll
                for (int t = 0; t < TILES; ++t) tie(b[t]);
            };
            static_assert(NA + TILES <= 15, "lgkmcnt range");
            auto compute_chunk = [&](const unsigned wslot, const int s0, const int ns) {
                // K-slot offsets of the chunk's steps -> registers (steps past the end repeat the last one; their MFMAs are skipped)
                unsigned toff[CONV_CH];
#pragma unroll
                for (int j = 0; j < CONV_CH; ++j)
                    asm volatile("ds_read_b32 %0, %1" : "=v"(toff[j]) : "v"(lds_tab + (unsigned)(((s0 + min(j, ns - 1)) * 4 + q) * 4)));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < CONV_CH; ++j) asm volatile("" : "+v"(toff[j]));
                const unsigned wa = wslot + (unsigned)lane * 16;
                auto load_set = [&](auto jc, uint4 (&a)[NA], uint4 (&b)[TILES]) {
                    constexpr int j = decltype(jc)::value;              // step within the chunk (steps >= ns re-read step ns - 1 ...
                    const unsigned wj = wa + (unsigned)min(j, ns - 1) * (NA * 1024);   // ... so that every read hits landed data)
#pragma unroll
                    for (int i = 0; i < NA; ++i) {
                        u32x4 v;
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(wj), "n"(i * 1024));
                        a[i] = make_uint4(v[0], v[1], v[2], v[3]);
                    }
#pragma unroll
                    for (int t = 0; t < TILES; ++t) b[t] = ld128(pbaddr[t] + toff[j]);
                };
                uint4 a0[NA], b0[TILES], a1[NA], b1[TILES];
                load_set(std::integral_constant<int, 0>{}, a0, b0);
#define RV_RING_PAIR(J)                                                                     \
                load_set(std::integral_constant<int, (J) + 1>{}, a1, b1);                   \
                wait_set(a0, b0);                                                           \
                __builtin_amdgcn_sched_barrier(0);                                          \
                if ((J) < ns) mfma_step(a0, b0);                                            \
                load_set(std::integral_constant<int, (J) + 2>{}, a0, b0);                   \
                wait_set(a1, b1);                                                           \
                __builtin_amdgcn_sched_barrier(0);                                          \
                if ((J) + 1 < ns) mfma_step(a1, b1);
                RV_RING_PAIR(0) RV_RING_PAIR(2) RV_RING_PAIR(4)
#undef RV_RING_PAIR
                static_assert(CONV_CH == 8, "the ring loop is unrolled for 8 K-steps per chunk");
                // last pair: NO read past the chunk's last step.  An asm read whose result is never used is a dead definition to
                // the register allocator -- it reuses the registers at once, and the LDS data landing later overwrites whatever
                // lives there by then (first version of this loop: corrupted the last step's B fragments of MT = 1 kernels).
                load_set(std::integral_constant<int, 7>{}, a1, b1);
                wait_set(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (6 < ns) mfma_step(a0, b0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < NA; ++i) tie(a1[i]);
#pragma unroll
                for (int t = 0; t < TILES; ++t) tie(b1[t]);
                __builtin_amdgcn_sched_barrier(0);
                if (7 < ns) mfma_step(a1, b1);
            };
            const int n_chunks = (p.S + CONV_CH - 1) / CONV_CH;
            zero_acc();
            int slot_r = 0;
            for (int c = 0; c < n_chunks; ++c) {
                const int after = min(NS - 2, n_chunks - 1 - c);       // chunks this wave has issued after chunk c
                if (after <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPW) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS stores (table, tile staging)
                __builtin_amdgcn_s_barrier();                          // chunk c landed for every wave; slot of chunk c - 1 is free
                asm volatile("" ::: "memory");
                if (c + NS - 1 < n_chunks) { issue(c + NS - 1, slot_w); slot_w = slot_w + 1 == NS ? 0 : slot_w + 1; }
                compute_chunk(lds_wl + (unsigned)slot_r * SLOT, c * CONV_CH, min(CONV_CH, p.S - c * CONV_CH));
                slot_r = slot_r + 1 == NS ? 0 : slot_r + 1;
            }
        } else {
            // Weight chunks: chunk c+1 is fetched into registers while chunk c feeds the MFMAs and parked afterwards.
            constexpr int WPT = (CONV_CH * MT * WFR * 64) / NT;          // uint4 per thread per chunk (4 * MT * WFR with 4 waves)
            const int n_chunks = (p.S + CONV_CH - 1) / CONV_CH;
            // named registers, not an array: hipcc keeps a prefetch *array* in scratch memory here (scratch_store right
            // behind every load), which serialises the whole prefetch
            static_assert(WPT <= 12, "prefetch register set");
            uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11;
    #define RV_W_ALL(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11)
    #define RV_W_LOAD(k) if constexpr (WPT > k) w##k = g[min(tid + k * NT, n16n - 1)];
    #define RV_W_STORE(k) if constexpr (WPT > k) { if (tid + k * NT < n16n) d[tid + k * NT] = w##k; }
            {
                const int n16 = min(CONV_CH, p.S) * MT * WFR * 64;
                uint4* d = reinterpret_cast<uint4*>(wl);
                for (int i = tid; i < n16; i += NT) d[i] = wsrc[i];      // chunk 0, issued together with the tile staging
            }
            zero_acc();
            __syncthreads();
            for (int c = 0; c < n_chunks; ++c) {
                const int s0 = c * CONV_CH;
                const int ns = min(CONV_CH, p.S - s0);
                const bool has_next = (c + 1 < n_chunks);
                // the prefetch is unconditional (the last iteration re-reads its own chunk and drops it) so that the
                // prefetch registers have one definition per iteration: a conditional definition makes hipcc copy them
                // right behind the loads, i.e. wait for every load before the MFMA loop
                const int sn = has_next ? s0 + CONV_CH : s0;
                const int n16n = min(CONV_CH, p.S - sn) * MT * WFR * 64;
                {
                    const uint4* g = wsrc + (size_t)sn * MT * WFR * 64;
                    RV_W_ALL(RV_W_LOAD)
                }
                compute_steps(wl, s0, ns);
                if (has_next) {
                    __syncthreads();                                   // everyone is done reading the chunk buffer
                    uint4* d = reinterpret_cast<uint4*>(wl);
                    RV_W_ALL(RV_W_STORE)
                    __syncthreads();
                }
            }
        }
        epilogue(ty0, tx0, tid);
    }
}

static int g_wg_cap = 0;
extern "C" int refvsr_set_conv_workgroup_cap(int cap) {
    g_wg_cap = cap > 0 ? cap : 0;
    return 0;
}

extern "C" int refvsr_kslot(int ty, int tx, int cg, int ksize, int ncg) { return rv_kslot(ty, tx, cg, ksize, ncg); }
extern "C" int refvsr_ksteps(int ksize, int ncg) { return rv_ksteps(ksize, ncg); }

// RESIDENT kernels launch only as many workgroups as the chip holds at once (occupancy x CUs, a multiple of 8 for
// the XCD banding) and walk the tiles; the others launch one workgroup per tile.
template <int MT, int TILES, bool F32, bool GATHER, bool RESIDENT, int EPI = 0, int NW = 4, bool HI1 = false>
static int launch_conv(ConvArgs& a, int nz, size_t lds, hipStream_t st) {
    // per device: the dynamic-LDS attribute and the occupancy table (a process may drive several GPUs)
    static bool attr_done[RV_MAX_DEVICES] = {};
    const int dev = rv_device();
    if (!attr_done[dev]) {
        RV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<MT, TILES, F32, GATHER, RESIDENT, EPI, NW, HI1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[dev] = true;
    }
    int gx = a.n_xy;
    if (RESIDENT) {
        static size_t occ_lds[RV_MAX_DEVICES][4] = {};
        static int occ_val[RV_MAX_DEVICES][4] = {};
        static int slot[RV_MAX_DEVICES] = {};
        int occ = 0;
        for (int i = 0; i < 4; ++i)
            if (occ_lds[dev][i] == lds && occ_val[dev][i] > 0) occ = occ_val[dev][i];
        if (occ == 0) {
            RV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_mfma_kernel<MT, TILES, F32, GATHER, RESIDENT, EPI, NW, HI1>, NW * 64, lds));
            if (occ < 1) occ = 1;
            occ_lds[dev][slot[dev] & 3] = lds; occ_val[dev][slot[dev] & 3] = occ; ++slot[dev];
        }
        int cap = (rv_stream_cus(st) * occ / (nz * (a.batch > 1 ? a.batch : 1))) & ~7;
        if (cap < 8) cap = 8;
        if (g_wg_cap > 0) cap = g_wg_cap;                          // refvsr_set_conv_workgroup_cap
        if (gx > cap) gx = cap;
    }
    a.grid = gx;
    hipLaunchKernelGGL((conv_mfma_kernel<MT, TILES, F32, GATHER, RESIDENT, EPI, NW, HI1>), dim3(gx, a.batch > 1 ? a.batch : 1, nz), dim3(NW * 64), lds, st, a);
    RV_LAUNCH_CHECK();
    return 0;
}

extern "C" int refvsr_conv_mfma(const RefvsrConv* d, void* stream) {
    RV_CHECK(d != nullptr, "conv: null descriptor");
    const bool f32 = d->f32 == 1;
    const bool hi1 = d->f32 == 2;                      // fp16 weights without the lo term (streamed kernels only)
    RV_CHECK(d->f32 >= 0 && d->f32 <= 2, "conv: weight mode (f32) must be 0, 1 or 2");
    const int cgrp = f32 ? 4 : 8;                      // channels per 16-byte group
    const int esz = f32 ? 4 : 2;
    RV_CHECK(d->src0 && d->c0 > 0 && d->c0 % cgrp == 0, "conv: src0/c0 invalid (c0=%d)", d->c0);
    RV_CHECK((d->src1 == nullptr) == (d->c1 == 0) && d->c1 % cgrp == 0, "conv: src1/c1 invalid (c1=%d)", d->c1);
    RV_CHECK(d->ksize >= 1 && d->ksize <= 7 && d->stride >= 1 && d->pad >= 0, "conv: bad geometry");
    RV_CHECK(d->h_in > 0 && d->w_in > 0 && d->h_out > 0 && d->w_out > 0, "conv: bad sizes");
    RV_CHECK(d->wpack && d->bias && d->out, "conv: null weights/bias/out");
    RV_CHECK(d->mt_per_block >= 1 && d->mt_per_block <= 3, "conv: mt_per_block must be 1..3");
    RV_CHECK(d->cout >= 1, "conv: cout");
    if (d->out_mode != REFVSR_OUT_PLANAR32) {
        RV_CHECK(d->cout % 4 == 0 && d->out_c % 4 == 0, "conv: nhwc16 output needs cout %% 4 == 0");
        RV_CHECK(d->res_planar == nullptr, "conv: res_planar only with planar output");
    }
    if (d->out_mode == REFVSR_OUT_NHWC16_SHUFFLE2)
        RV_CHECK(d->cout % 16 == 0 && !d->mul && !d->res && !f32 && d->out_c >= d->cout / 4 && d->out_c - d->cout / 4 <= 4,
                 "conv: pixel-shuffle output constraints");
    RV_CHECK(refvsr_init() == 0, "init failed");

    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = (const unsigned char*)d->src0; a.src1 = (const unsigned char*)d->src1;
    a.c0 = d->c0; a.c1 = d->c1;
    a.pixb0 = d->c0 * esz; a.pixb1 = d->c1 * esz;
    a.ncg0 = d->c0 / cgrp; a.ncg = (d->c0 + d->c1) / cgrp;
    a.ps = a.ncg | 1;
    a.h_in = d->h_in; a.w_in = d->w_in; a.h_out = d->h_out; a.w_out = d->w_out;
    a.ks = d->ksize; a.stride = d->stride; a.pad = d->pad;
    a.G = d->ksize * d->ksize * a.ncg;
    a.S = rv_ksteps(d->ksize, a.ncg);
    RV_CHECK(a.S == d->ksteps, "conv: ksteps mismatch (descriptor %d, geometry %d)", d->ksteps, a.S);
    a.inv_ncg = 1.0f / (float)a.ncg;
    a.wpack = (const uint4*)d->wpack; a.bias = d->bias; a.cout = d->cout;
    a.act_slope = d->act_slope; a.post_slope = d->post_slope;
    a.mul = (const unsigned char*)d->mul; a.mul_c = d->mul_c;
    a.res = (const unsigned char*)d->res; a.res_c = d->res_c;
    a.out_mode = d->out_mode; a.out = d->out; a.out_c = d->out_c;
    a.res_planar = d->res_planar; a.add_const = d->add_const;
    a.clamp_lo = d->clamp_lo; a.clamp_hi = d->clamp_hi;
    RV_CHECK(d->batch >= 0 && d->batch <= 65535, "conv: batch out of range (%d)", d->batch);
    a.batch = d->batch > 1 ? d->batch : 1;
    if (a.batch > 1) {
        RV_CHECK(!d->mul && !d->res, "conv: batch > 1 takes no mul / res operands");
        RV_CHECK(d->bs_src0 % 16 == 0 && d->bs_src1 % 16 == 0 && d->bs_out % 8 == 0 && d->bs_res_planar % 4 == 0, "conv: batch strides must keep the maps aligned");
        RV_CHECK(d->bs_src0 > 0 && d->bs_out > 0 && (!d->src1 || d->bs_src1 > 0) && (!d->res_planar || d->bs_res_planar > 0), "conv: batch strides missing");
        a.bs_src0 = d->bs_src0; a.bs_src1 = d->bs_src1; a.bs_out = d->bs_out; a.bs_res_planar = d->bs_res_planar;
    }

    const int MT = d->mt_per_block;
    const int n_mt = (d->cout + 15) / 16;
    const int nz = (n_mt + MT - 1) / MT;
    a.tab_bytes = ((a.S * 4 * 4 + 15) / 16) * 16;
    const int wfr_kb = MT * ((f32 || hi1) ? 1 : 2) * 1024;   // bytes of weight fragments per K-step
    static const bool no_resident = getenv("REFVSR_CONV_NO_PERSIST") != nullptr;   // A/B knob, read once
    const size_t LDS_MAX = 160 * 1024;

    int tiles = 4;
    size_t lds = 0;
    auto tile_bytes = [&](int tl) {
        a.LH = (tl * 2 - 1) * a.stride + a.ks;
        a.LW = (CONV_TW - 1) * a.stride + a.ks;
        return (size_t)a.LH * a.LW * a.ps * 16;
    };
    // RESIDENT: whole weight set in LDS, persistent workgroups with a register-prefetched input tile.  Needs few enough
    // K-steps, a tile that fits the prefetch registers, and LDS for >= 2 workgroups per CU (8 x 32 pixels preferred,
    // 4 x 32 when that buys a second / third workgroup).
    bool resident = false, one_wg = false;             // one_wg: LDS admits a single workgroup per CU
    bool w16 = false;                                  // 16 x 32 tile, 16 waves
    static const int res_max = getenv("REFVSR_CONV_RES_MAX") ? atoi(getenv("REFVSR_CONV_RES_MAX")) : CONV_RES_MAX;   // A/B knob
    if (!no_resident && !hi1 && a.S <= res_max && a.S <= CONV_RES_MAX) {
        int best_wg = 0;
        static const int force_tiles = getenv("REFVSR_CONV_TILES") ? atoi(getenv("REFVSR_CONV_TILES")) : 0;   // A/B knob: 2 | 4
        for (int tl = 4; tl >= 2; tl -= 2) {
            if (force_tiles && tl != force_tiles) continue;
            const size_t tb = tile_bytes(tl);
            const size_t need = (size_t)a.tab_bytes + (size_t)a.S * wfr_kb + tb;
            const int chunks = a.LH * a.LW * a.ncg;
            const int wg = need <= LDS_MAX ? (int)(LDS_MAX / need) : 0;
            if (chunks > CONV_XPF * 256 || wg == 0) continue;
            if (a.LH > 31 || a.LW > 127 || a.ncg > 127) continue;   // packed chunk descriptor of the tile staging (r:5, c:7, cg:7 + 7 bits)
            if (best_wg == 0 || (best_wg < 2 && wg > best_wg)) { best_wg = wg; tiles = tl; lds = need; resident = true; }
        }
        if (resident) { a.wl_bytes = a.S * wfr_kb; tile_bytes(tiles); one_wg = best_wg == 1; }
        // One workgroup per CU (the C = 48 weight sets): a 16 x 32 tile walked by SIXTEEN waves (two pixel groups each) keeps
        // four waves per SIMD next to the 84 KB weight set and halves the halo; 8 waves on 8 x 32 where that does not fit.
        // (same box, frames/s: RefVSR_MFID 60.6 [4 waves] / 68.9 [8 x 2] / 70.1 [8 x 4] / 72.2 [16 x 2]; MFID_8K 1080p 5.04 / 5.80 / 6.24 / 6.25)
        static const bool no_w16 = getenv("REFVSR_CONV_NO_W16") != nullptr;          // A/B knob, read once
        if (resident && one_wg && tiles == 4 && !f32 && !no_w16 && (MT == 2 || MT == 3)) {
            const size_t tb = tile_bytes(8);
            const size_t need = (size_t)a.tab_bytes + (size_t)a.S * wfr_kb + tb;
            const int chunks = a.LH * a.LW * a.ncg;
            if (need <= LDS_MAX && chunks <= 4096 && a.LH <= 31 && a.LW <= 127) { tiles = 8; lds = need; w16 = true; }
            else tile_bytes(tiles);
        }
    }
    if (!resident) {
        // streamed weights: a ring of 2..4 LDS slots of CONV_CH K-steps each next to the staged input tile.  Large maps prefer a
        // footprint that admits two workgroups per CU; 8 x 32 pixels if the staged input fits, else 4 x 32, else gather mode
        // (one slot, register-prefetched, B fragments from global memory)
        const size_t slot = (size_t)CONV_CH * wfr_kb;
        const int n_chunks = (a.S + CONV_CH - 1) / CONV_CH;
        auto ring_for = [&](int tl, size_t budget) {
            const size_t fixed = (size_t)a.tab_bytes + tile_bytes(tl);
            if (fixed + 2 * slot > budget) return 0;
            int ns = (int)((budget - fixed) / slot);
            if (ns > 4) ns = 4;
            if (ns > n_chunks + 1) ns = n_chunks + 1 > 2 ? n_chunks + 1 : 2;
            return ns;
        };
        static const int force_ring = getenv("REFVSR_CONV_RING") ? atoi(getenv("REFVSR_CONV_RING")) : 0;   // A/B knob: 2 | 3 | 4
        const bool big = (long long)d->h_out * d->w_out > 64 * 1024;
        int ns = 0;
        tiles = 4;
        if (big) ns = ring_for(4, LDS_MAX / 2);
        if (!ns && big) { tiles = 2; ns = ring_for(2, LDS_MAX / 2); }
        if (!ns) { tiles = 4; ns = ring_for(4, LDS_MAX); }
        if (!ns) { tiles = 2; ns = ring_for(2, LDS_MAX); }
        if (ns) {
            if (force_ring >= 2 && force_ring < ns) ns = force_ring;
            a.ring = ns;
            a.wl_bytes = (int)(ns * slot);
            lds = (size_t)a.tab_bytes + (size_t)a.wl_bytes + tile_bytes(tiles);
        } else {                                       // strided predictor convs: gather B fragments from global memory
            a.gather = 1;
            a.ring = 1;
            tiles = 4;
            a.LH = a.LW = 0;
            a.wl_bytes = (int)slot;
            lds = (size_t)a.tab_bytes + (size_t)a.wl_bytes;
            RV_CHECK(d->h_in < 60000 && d->w_in < 60000, "conv: frame too large for gather-mode coordinates");
        }
        one_wg = !a.gather && lds > LDS_MAX / 2;
    }
    // one workgroup per CU: eight waves on the same 8 x 32 tile (two pixel groups per wave) keep two waves per SIMD
    static const bool no_nw8 = getenv("REFVSR_CONV_NO_NW8") != nullptr;             // A/B knob, read once
    // ... and eight waves (two pixel groups each) on the other 8 x 32 fp16 tiles as well, unless the map is large: 2 workgroups x
    // 8 waves = 4 waves per SIMD instead of 3 x 4 = 3 hides more latency (24->24 at 270p 9.7 -> 9.2 us, at 540p 24.3 -> 23.0 us,
    // same box 156.7 -> 159.3 frames/s on RefVSR_small); at 1080p (4080 tiles, 8 per workgroup) the better weight-fragment
    // reuse of four pixel groups per wave wins (75.5 vs 78.5 us).  (Forcing <= 80 VGPRs for 6 waves per SIMD spills.)
    const int n_tiles8 = rv_cdiv(d->w_out, CONV_TW) * rv_cdiv(d->h_out, 8);
    const bool nw8 = tiles == 4 && !f32 && !a.gather && !no_nw8 && (one_wg || n_tiles8 <= 2048);
    static const bool no_prefetch = getenv("REFVSR_CONV_NO_PREFETCH") != nullptr;   // A/B knob, read once
    a.prefetch = no_prefetch ? 0 : 1;
    a.tiles_x = rv_cdiv(d->w_out, CONV_TW);
    a.n_xy = a.tiles_x * rv_cdiv(d->h_out, tiles * 2);
    hipStream_t st = (hipStream_t)stream;
    if (a.gather) {                                // only the fp16 strided predictors need it
        RV_CHECK(!f32 && !hi1, "conv: gather mode is built for the fp16 hi+lo path only");
        if (MT == 1) return launch_conv<1, 4, false, true, false>(a, nz, lds, st);
        if (MT == 2) return launch_conv<2, 4, false, true, false>(a, nz, lds, st);
        return launch_conv<3, 4, false, true, false>(a, nz, lds, st);
    }
    if (hi1) {                                     // SPyNet's streamed 7x7 convs (Engine.flow): half the weight stream, half the MFMAs
        RV_CHECK(!a.gather && MT <= 2, "conv: single-fp16 weights are built for the streamed stride-1 convs (MT=%d)", MT);
        const bool nw8h = tiles == 4 && !no_nw8 && (one_wg || n_tiles8 <= 2048);
        if (nw8h) return MT == 1 ? launch_conv<1, 2, false, false, false, 0, 8, true>(a, nz, lds, st)
                                 : launch_conv<2, 2, false, false, false, 0, 8, true>(a, nz, lds, st);
        if (tiles == 4) return MT == 1 ? launch_conv<1, 4, false, false, false, 0, 4, true>(a, nz, lds, st)
                                       : launch_conv<2, 4, false, false, false, 0, 4, true>(a, nz, lds, st);
        return MT == 1 ? launch_conv<1, 2, false, false, false, 0, 4, true>(a, nz, lds, st)
                       : launch_conv<2, 2, false, false, false, 0, 4, true>(a, nz, lds, st);
    }
    // lean epilogue: fp16 HWC output, slopes in [0, 1], maps addressable with 32-bit element offsets, tile coordinates in
    // the packed chunk descriptor's range
    static const bool no_lean = getenv("REFVSR_CONV_NO_LEAN_EPI") != nullptr;      // A/B knob, read once
    const bool lean = resident && !f32 && !no_lean && d->out_mode == REFVSR_OUT_NHWC16 && d->act_slope >= 0.f && d->act_slope <= 1.f &&
                      d->post_slope >= 0.f && d->post_slope <= 1.f &&
                      (long long)d->h_out * d->w_out * (long long)(d->out_c > d->mul_c ? (d->out_c > d->res_c ? d->out_c : d->res_c)
                                                                                       : (d->mul_c > d->res_c ? d->mul_c : d->res_c)) < (1ll << 31);
#define RV_CONV_CASE(M, T)                                                                                \
    if (MT == M && tiles == T) {                                                                          \
        if (f32) return resident ? launch_conv<M, T, true, false, true>(a, nz, lds, st)                   \
                                 : launch_conv<M, T, true, false, false>(a, nz, lds, st);                 \
        if (lean) return launch_conv<M, T, false, false, true, 1>(a, nz, lds, st);                        \
        return resident ? launch_conv<M, T, false, false, true>(a, nz, lds, st)                           \
                        : launch_conv<M, T, false, false, false>(a, nz, lds, st);                         \
    }
    if (w16) {
        if (MT == 3) return lean ? launch_conv<3, 2, false, false, true, 1, 16>(a, nz, lds, st) : launch_conv<3, 2, false, false, true, 0, 16>(a, nz, lds, st);
        return lean ? launch_conv<2, 2, false, false, true, 1, 16>(a, nz, lds, st) : launch_conv<2, 2, false, false, true, 0, 16>(a, nz, lds, st);
    }
    if (nw8) {
#define RV_CONV_CASE8(M)                                                                                  \
        if (MT == M) {                                                                                    \
            if (lean) return launch_conv<M, 2, false, false, true, 1, 8>(a, nz, lds, st);                 \
            return resident ? launch_conv<M, 2, false, false, true, 0, 8>(a, nz, lds, st)                 \
                            : launch_conv<M, 2, false, false, false, 0, 8>(a, nz, lds, st);               \
        }
        RV_CONV_CASE8(1) RV_CONV_CASE8(2) RV_CONV_CASE8(3)
#undef RV_CONV_CASE8
    }
    // the exact-fp32 convs (VGG head of the matching) on 4 x 32 tiles: eight waves with one pixel group each (fp32 MFMAs are
    // slow enough that the lost fragment reuse costs nothing: 64->64 at 270p 115 -> 105 us)
    if (!no_nw8 && f32 && tiles == 2 && !a.gather && (MT == 1 || MT == 2)) {
        if (MT == 1) return resident ? launch_conv<1, 1, true, false, true, 0, 8>(a, nz, lds, st) : launch_conv<1, 1, true, false, false, 0, 8>(a, nz, lds, st);
        return resident ? launch_conv<2, 1, true, false, true, 0, 8>(a, nz, lds, st) : launch_conv<2, 1, true, false, false, 0, 8>(a, nz, lds, st);
    }
    RV_CONV_CASE(1, 2) RV_CONV_CASE(1, 4)
    RV_CONV_CASE(2, 2) RV_CONV_CASE(2, 4)
    RV_CONV_CASE(3, 2) RV_CONV_CASE(3, 4)
#undef RV_CONV_CASE
    refvsr_set_error("conv: no kernel for MT=%d tiles=%d", MT, tiles);
    return 1;
}
