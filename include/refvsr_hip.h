/*
 * refvsr_hip.h -- C-ABI of the MI355X (gfx950) RefVSR inference hot path.
 *
 * The reference (codeslake/RefVSR) is pure Python + ATen ops; it has no FFI of its own.  The entry
 * points below are what a ctypes/cffi binding of the hot path binds instead of the ATen calls made
 * by models/archs/RefVSR.py:Network.forward (paths relative to the reference tree).  Every function
 * cites the reference code it replaces.  Plain pointers (device memory), ints, floats; no torch
 * types.  All kernels are enqueued on `stream` (a hipStream_t passed as void*) and return
 * immediately; 0 = success, non-zero = error (message via refvsr_last_error()).
 *
 * Device data layouts
 *   planar  : float32 [C][H][W]                  (frames, flows, confidence maps, VGG features)
 *   nhwc16  : _Float16 [H][W][Cs], Cs % 8 == 0   (all C-channel feature maps; "HWC tiles")
 * Batch size is 1 at this level (the host loops over n, as the reference's eval does).
 */
#ifndef REFVSR_HIP_H
#define REFVSR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REFVSR_ABI_VERSION 14  /* 2: K-block order of packed conv weights (refvsr_amd/packing.py:kslot);
                                  3: exact matching (match_refine flagging, match_exact), lean ResBlock;
                                  4: hi + lo patch rows (match_patches rows_lo), split-fp16 match_exact;
                                  5: compile-time-specialised 24-channel ResBlock (resblock24 blob);
                                  6: RefvsrConv.warp_* (inter-frame warp fused into the conv's tile staging);
                                  7: refvsr_conv24 / refvsr_conv48 (compile-time-specialised 3x3 convs);
                                  8: RefvsrConv.f32 = 2 (plain fp16 weights in the streamed convs);
                                  9: refvsr_conv_shuffle2 (compile-time-specialised C -> 4 C conv + pixel shuffle), refvsr_conv32;
                                  10: RefvsrConv.batch (one launch over several images that share the weights),
                                      refvsr_spynet_level_input_batch, refvsr_conf_alpha, refvsr_warp_nhwc16's flow_scale;
                                  11: multi-map launches (REFVSR_MAX_MAPS maps of one geometry behind one launch and one weight fill):
                                      refvsr_resblock24_chain_batch, refvsr_conv24_batch, refvsr_conv_shuffle2_batch,
                                      refvsr_conf_alpha_batch, refvsr_warp_nhwc16_batch, refvsr_warp_nhwc16_up2_batch,
                                      refvsr_warp_planar_batch; RefvsrConv.warp_* (ABI 6: the inter-frame warp fused into a conv's tile
                                      staging) REMOVED: bit-identical to warp + conv but measured slower in every configuration
                                      (169 vs 176 frames/s, profiles/r03_fused_warp_ab.txt);
                                  12: refvsr_resblock48_chain_batch (the multi-map form of the 48-channel block);
                                  13: CU partitions: refvsr_stream_create_cu_range / refvsr_stream_set_cu_budget /
                                      refvsr_stream_destroy / refvsr_num_cus;
                                  14: result formats of the output head (REFVSR_RESULT_*): refvsr_conv_last_fmt,
                                      refvsr_conv_hr_last_fmt, refvsr_convert_result */

int refvsr_abi_version(void);
/* REFVSR_MAX_MAPS of the built library (a value, not a status): bindings check their own copy against it at load time. */
int refvsr_max_maps(void);
const char* refvsr_last_error(void);
/* One-time per-process setup (raises dynamic-LDS limits).  Called lazily by every entry point. */
int refvsr_init(void);

/* CU partitions (ABI 13; no reference counterpart -- the reference leaves kernel placement to the CUDA runtime).
 * The path runs three internal streams (per-frame preparation, forward branch, backward branch: models/archs/RefVSR.py:196-204 is
 * independent of :211-238 of the previous frames); kernels of two streams only run side by side if both fit a CU, which the
 * LDS-heavy kernels of this library never do.  A stream created here executes on CUs [first_cu, first_cu + n_cus) only (both
 * multiples of 8: an equal share of every XCD) and the persistent launchers of this library size their grids for n_cus CUs on it.
 * refvsr_stream_set_cu_budget overrides the CU count the launchers assume on any stream (0 = forget the stream).
 * Results never depend on the partition.  refvsr_num_cus returns the CU count of the current device (a value, not a status). */
int refvsr_num_cus(void);
int refvsr_stream_create_cu_range(int first_cu, int n_cus, void** stream);
int refvsr_stream_set_cu_budget(void* stream, int n_cus);
int refvsr_stream_destroy(void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolution (implicit GEMM on v_mfma_f32_16x16x32_f16, fp32 accumulate).
 * Replaces every nn.Conv2d on C-channel maps: ResidualBlocksWithInputConv (RefVSR.py:327-360),
 * ResBlock/ResList/BasicBlock (RefVSR_/common.py:25-109), PixelShufflePack (mmedit upsample.py:36-51),
 * SPyNetBasicModule (SPyNet.py:142-202), AlignedConv2d.conv1/p_conv (RefVSR_/alignment.py:18-24),
 * fusion_UP/conv_hr/conv_last (RefVSR.py:87-92), with the surrounding elementwise work fused:
 * torch.cat of two inputs, bias, (Leaky)ReLU, `x + alpha*y` (RefVSR.py:131,143), residual adds,
 * pixel_shuffle, `+ base` and clamp (RefVSR.py:118,297).
 * ------------------------------------------------------------------------------------------ */
enum { REFVSR_OUT_NHWC16 = 0, REFVSR_OUT_NHWC16_SHUFFLE2 = 1, REFVSR_OUT_PLANAR32 = 2 };

typedef struct RefvsrConv {
    const void* src0; int c0;          /* nhwc16 input, channel stride c0 (multiple of 8)            */
    const void* src1; int c1;          /* optional 2nd input concatenated after src0 (or NULL, 0)    */
    int h_in, w_in;                    /* input spatial size                                         */
    int h_out, w_out;                  /* conv output spatial size (before pixel shuffle)            */
    int ksize, stride, pad;
    const void* wpack;                 /* packed fp16 weights, see refvsr_amd/packing.py             */
    const float* bias;                 /* fp32 [n_mtiles*16], packed row order                        */
    int cout;                          /* valid output rows (packed row order)                        */
    int mt_per_block;                  /* 1, 2 or 3 sixteen-row tiles handled per block               */
    int ksteps;                        /* number of 32-deep K steps in wpack                          */
    float act_slope;                   /* y<0 ? y*slope : y   (1 = linear, 0 = ReLU)                  */
    const void* mul; int mul_c;        /* optional nhwc16 multiplier (alpha) and its channel stride   */
    const void* res; int res_c;        /* optional nhwc16 residual added after mul                    */
    float post_slope;                  /* activation applied after the residual add (1 = none)        */
    int out_mode;                      /* REFVSR_OUT_*                                                */
    void* out; int out_c;              /* nhwc16 modes: channel stride of out                          */
    const float* res_planar;           /* PLANAR32: optional planar fp32 residual [cout][h][w]        */
    int f32;                           /* 0: nhwc16 maps, fp16 hi+lo weights on 16x16x32 f16 MFMA;
                                          1: the same maps in fp32 ("nhwc32", channel stride % 4 == 0), fp32 weights,
                                             exact fp32 products on v_mfma_f32_16x16x4_f32 (src/mul/res/out all fp32);
                                          2 (ABI 8): like 0 with plain fp16 weights (no lo term; wpack [nz][S][MT][1][64][8]):
                                             half the weight stream and half the MFMAs -- streamed convs only (more than 16
                                             K-steps, stride 1, mt_per_block <= 2); used for SPyNet's 7x7 convs, whose
                                             contribution to the end-to-end error budget is measured in DESIGN.md section 2 */
    float add_const;                   /* PLANAR32: constant added after the residual                 */
    float clamp_lo, clamp_hi;          /* PLANAR32: clamp when clamp_lo < clamp_hi                    */
    /* Batch (ABI 10): batch > 1 runs the same conv over `batch` images in ONE launch (blockIdx.y = image); image b reads
     * src0 + b * bs_src0 (src1 + b * bs_src1) and writes out + b * bs_out, res_planar + b * bs_res_planar (byte strides).
     * mul / res must be NULL then.  batch = 0 or 1: a single image (strides ignored).  Used for the two SPyNet
     * flows a frame needs (SPyNet.py:49-104 is called per pair by the reference; same weights, same shapes): the coarse
     * pyramid levels are launches of 2-36 workgroups, two images per launch fill twice the CUs for the same latency. */
    int batch; size_t bs_src0, bs_src1, bs_out, bs_res_planar;
} RefvsrConv;

int refvsr_conv_mfma(const RefvsrConv* d, void* stream);
/* Tuning / test knob (no reference counterpart): upper bound on the workgroups one single-chunk conv launches; each
 * workgroup walks the remaining pixel tiles.  0 = automatic (occupancy x CUs).  Results do not depend on it. */
int refvsr_set_conv_workgroup_cap(int cap);
/* Packing contract of `wpack` (host-side helpers, no GPU needed): K-slot of K-block (ty, tx, cg) of a ks x ks conv over
 * ncg 16-byte channel groups, and the number of 4-slot K-steps; refvsr_amd/packing.py:kslot/ksteps are the same closed
 * forms and tests/test_capi.py pins the two against each other.  Return the value (>= 0), not a status. */
int refvsr_kslot(int ty, int tx, int cg, int ksize, int ncg);
int refvsr_ksteps(int ksize, int ncg);

/* Fused residual block  out = post( x + conv2( act( conv1(x) ) ) ),  3x3, C -> C, stride 1, in ONE launch (the intermediate map
 * lives in LDS): ResidualBlockNoBN (mmedit sr_backbone_utils.py:42-97) and ResBlock (RefVSR_/common.py:25-39).  w1 / w2: fp16 hi + lo
 * packed weights of the two convs, b1 / b2 packed biases.  Runtime-generic kernel for C in {8, 16, 24, 32} (refvsr_resblock_lean_fits):
 * 8 waves on an 8 x 32 tile, both weight sets + one activation tile in LDS (the intermediate map overwrites the input tile): two
 * workgroups per CU, so one's epilogue runs under the other's MFMAs.  Same arithmetic as two refvsr_conv_mfma launches up to fp32 summation order.  Slopes must
 * lie in [0, 1].  (Round 1's 16 x 32-tile kernel, refvsr_resblock_mfma, was the reference of the bit-identity tests until round 4 and
 * is gone: every shape it ran is covered by this kernel and by refvsr_resblock24_chain.) */
int refvsr_resblock_lean_fits(int c);
/* n fused blocks behind one call (n launches, intermediates ping-pong between scratch0 / scratch1: each [h][w][c] fp16,
 * scratch0 needed for n >= 2, scratch1 for n >= 3; src, scratch*, out pairwise distinct).  w1 / b1 / w2 / b2: host arrays of
 * n device pointers.  Same results as n calls of refvsr_resblock_lean; saves n - 1 FFI crossings and n - 2 allocations. */
int refvsr_resblock_chain(const void* src, int c, int h, int w, int n, const void* const* w1, const float* const* b1,
                          const void* const* w2, const float* const* b2, int ksteps, float act_slope, float post_slope,
                          void* scratch0, void* scratch1, void* out, void* stream);
/* Tuning knob: waves per workgroup of the lean kernel, 8 (default: half the tiles per wave, <= 128 VGPRs, four waves per
 * SIMD) or 4.  Results do not depend on it. */
int refvsr_set_resblock_waves(int waves);
int refvsr_resblock_lean(const void* src, int c, int h, int w, const void* w1, const float* b1,
                         const void* w2, const float* b2, int ksteps, float act_slope, float post_slope,
                         void* out, void* stream);
/* The 24-channel fused block of RefVSR_small (mid_channels = 24: configs/config_RefVSR_small_*.py) with every geometry
 * constant fixed at compile time -- ResidualBlockNoBN (mmedit sr_backbone_utils.py:42-97, act_slope 0 = ReLU) and ResBlock
 * (RefVSR_/common.py:25-39, act_slope 0.2), out = x + conv2(act(conv1 x)), n blocks behind one call like
 * refvsr_resblock_chain.  Block i's parameters are ONE blob of REFVSR_RESBLOCK24_BLOB_BYTES at blobs + i * blob_stride
 * (device memory, 16-byte aligned): [conv1: 7 K-steps x 3 fragments x 64 lanes x 8 halfs][conv2: same][b1: 32 floats,
 * 24..31 = 0][b2: 32 floats]; fragments 0 / 1 = hi / lo halves of output channels 0-15, fragment 2 = rows 0-7 hi, rows 8-15
 * lo of channels 16-23; lane l = (q = l >> 4, row r = l & 15) holds K-block refvsr_resblock24_kblock(s, q) of row r
 * (refvsr_amd/packing.py:pack_resblock24 builds it).  Same arithmetic as refvsr_resblock_lean up to fp32 summation order. */
#define REFVSR_RESBLOCK24_BLOB_BYTES 43264
int refvsr_resblock24_chain(const void* src, int h, int w, int n, const void* blobs, size_t blob_stride, float act_slope,
                            void* scratch0, void* scratch1, void* out, void* stream);
/* Multi-map launches (ABI 11).  The propagation branches of CONSECUTIVE output frames are independent chains over identical
 * weights (the backward branch restarts from zeros in every window, RefVSR.py:211-238), and so are the n samples of a batch
 * (lrs [n,t,3,h,w], RefVSR.py:151): `batch` <= REFVSR_MAX_MAPS maps of one geometry run behind ONE launch per layer.  A workgroup
 * walks its share of the batch x tiles on one weight fill (at 270 x 480 a launch is one 8 x 32 tile per workgroup and a third of its
 * life is the fill).  Maps are given as HOST arrays of `batch` device pointers (they need not be contiguous: per-frame cached maps
 * enter the chains).  Map b of a batched call == the single-map call on map b, bit for bit (tests/test_gpu_ops.py). */
#define REFVSR_MAX_MAPS 4
/* refvsr_resblock24_chain over `batch` maps: src / out host arrays of batch pointers ([h][w][24] fp16 each), scratch0 / scratch1:
 * [batch][h][w][24] contiguous (needed as in the single-map call). */
int refvsr_resblock24_chain_batch(const void* const* src, int batch, int h, int w, int n, const void* blobs, size_t blob_stride,
                                  float act_slope, void* scratch0, void* scratch1, void* const* out, void* stream);
/* (ty << 16 | tx << 8 | cg) of K-block (K-step s = 0..6, quarter q = 0..3) of the blob's K order, -1 for the zero block. */
int refvsr_resblock24_kblock(int s, int q);
/* Tuning knob: workgroup shape of the 24-channel kernel.  0 (default): by map size -- 8 waves on 8 x 32-pixel tiles, or 16
 * waves on 16 x 32 tiles (one workgroup per CU) when the map has at least four 8 x 32 tiles per CU; 4 | 8 | 16 force a
 * shape.  Results do not depend on it (bit-identical). */
int refvsr_set_resblock24_waves(int waves);
/* The 48-channel fused block (ABI 10; mid_channels = 48: configs/config_RefVSR_{L1,MFID,MFID_8K}.py, 30 ResidualBlockNoBN per
 * branch + the ResBlocks of the ResLists): out = x + conv2(act(conv1 x)) in ONE launch per block.  One conv's 84 KB of hi + lo
 * fragments is resident at a time; the two sets swap per 8 x 32-pixel tile by LDS-DMA under the phases that do not read them
 * (csrc/resblock48.hip).  Block i's parameters are ONE blob of REFVSR_RESBLOCK48_BLOB_BYTES at blobs + i * blob_stride:
 * [conv1: 14 K-steps x 6 fragments x 64 lanes x 8 halfs][conv2: same][b1: 64 floats, 48.. = 0][b2: 64 floats] -- the fragment
 * parts of the two refvsr_conv48 blobs (K-blocks by refvsr_conv24_kblock(6, s, q)); refvsr_amd/packing.py:pack_resblock48.
 * Same arithmetic as two refvsr_conv48 launches (bit-identical: same K order, same fp16 rounding of the intermediate). */
#define REFVSR_RESBLOCK48_BLOB_BYTES 172544
int refvsr_resblock48_chain(const void* src, int h, int w, int n, const void* blobs, size_t blob_stride, float act_slope,
                            void* scratch0, void* scratch1, void* out, void* stream);
/* refvsr_resblock48_chain over `batch` <= REFVSR_MAX_MAPS maps of one geometry (ABI 12; the backward branches of consecutive output
 * frames of the mid_channels = 48 models, like refvsr_resblock24_chain_batch): ONE launch per block over all maps -- the launch's fixed
 * cost, the first 86 KB weight fill and the tail are shared (the two weight sets still swap per tile).  src / out: host arrays of
 * batch device pointers ([h][w][48] fp16 each), scratch0 / scratch1: [batch][h][w][48] contiguous.  Map b == the single-map call on
 * map b, bit for bit. */
int refvsr_resblock48_chain_batch(const void* const* src, int batch, int h, int w, int n, const void* blobs, size_t blob_stride,
                                  float act_slope, void* scratch0, void* scratch1, void* const* out, void* stream);
/* Tuning knob: how the 24-channel kernel stores its output tile.  0: 8-byte stores (one per lane and pixel group); 1: 16-byte
 * stores after a v_permlane16_swap exchange between neighbouring lane rows (half the store instructions; default).  Results do not
 * depend on it (bit-identical). */
int refvsr_set_resblock24_store(int mode);
/* 3x3 stride-1 pad-1 convolutions with 24 output channels on fp16 HWC maps, compile-time specialised like the block above
 * (csrc/conv24.hip): the ResList tails (RefVSR_/common.py:80-82), feat_fusion / conf_fusion / fusion_UP convs (RefVSR.py:47-62,87),
 * ref encoders, conv_hr and the input conv of ResidualBlocksWithInputConv (RefVSR.py:340-343) of the mid_channels = 24 family.
 * out = post( act(conv(cat[src0, src1]) + bias) * mul + res ); (c0, c1) in {(24,0), (16,0), (8,24), (24,24)} = channel strides of
 * the sources (src1 NULL when c1 = 0); out / mul / res: 24-channel maps [h][w][24] fp16 (mul, res optional).  blob: the conv's
 * parameters, refvsr_conv24_blob_bytes(c0, c1) bytes, 16-byte aligned: [S K-steps x 3 fragments x 64 lanes x 8 halfs][32 bias
 * floats]; fragment rows as in the resblock24 blob, lane (q, r) of K-step s holds K-block refvsr_conv24_kblock(ncg, s, q)
 * (ty << 16 | tx << 8 | cg over the (c0 + c1) / 8 channel groups of the concatenated sources; -1 = zero block);
 * refvsr_amd/packing.py:pack_conv24 builds it.  Same arithmetic as refvsr_conv_mfma up to fp32 summation order. */
int refvsr_conv24_supported(int c0, int c1);
int refvsr_conv24_blob_bytes(int c0, int c1);
int refvsr_conv24_kblock(int ncg, int s, int q);
int refvsr_conv24(const void* src0, int c0, const void* src1, int c1, int h, int w, const void* blob, float act_slope,
                  const void* mul, const void* res, float post_slope, void* out, void* stream);
/* refvsr_conv24 over `batch` maps (ABI 11; 24 output channels): every per-map operand is a host array of `batch` device pointers --
 * src0, src1 (NULL array when c1 = 0), mul / res (NULL array = none; else every entry non-NULL), out. */
int refvsr_conv24_batch(const void* const* src0, int c0, const void* const* src1, int c1, int batch, int h, int w, const void* blob,
                        float act_slope, const void* const* mul, const void* const* res, float post_slope, void* const* out,
                        void* stream);
/* The output head in ONE launch (ABI 10): RefVSR.py:92,118,288,297 --
 *   out = clamp( conv_last(src) + clamp(F.interpolate(lr_centre, scale_factor=scale, mode='bicubic'), 0, 1), 0, 1 )
 * -- planar fp32 [3][h][w].  src: fp16 HWC [h][w][c], c = 24 | 48; base_lr: planar fp32 [3][bh][bw], h / bh == w / bw = the SR factor;
 * blob: refvsr_conv_last_blob_bytes(c) bytes = [S K-steps x ONE fragment x 64 lanes x 8 halfs][32 bias floats], fragment rows
 * 0-2 = hi(W[r]), rows 8-10 = lo(W[r - 8]) (folded by the kernel), K-blocks by refvsr_conv24_kblock(c / 8, s, q);
 * refvsr_amd/packing.py:pack_conv_last.  The base map (refvsr_resize bicubic x4: 25 MB written and read back at 1080p) is
 * evaluated per output value instead.  Same arithmetic as refvsr_resize + refvsr_conv_mfma's planar mode up to the fp32
 * summation order of the conv. */
int refvsr_conv_last_supported(int c);
int refvsr_conv_last_blob_bytes(int c);
int refvsr_conv_last(const void* src, int c, int h, int w, const void* blob, const float* base_lr, int bh, int bw,
                     float* out, void* stream);
/* conv_hr AND the head in one launch (ABI 10; mid_channels = 24): RefVSR.py:91-92,116-118,288,297 --
 *   out = clamp( conv_last( lrelu_{act_slope}( conv_hr(src) ) ) + clamp(F.interpolate(lr_centre, bicubic), 0, 1), 0, 1 )
 * on the two-conv skeleton of refvsr_resblock24_chain: the intermediate HR map (100 MB at 1080 x 1920) stays in LDS.  src: fp16
 * HWC [h][w][24]; blob: REFVSR_RESBLOCK24_BLOB_BYTES in the block layout with conv1 = conv_hr and conv2's fragment slot (s, 0) =
 * [rows 0-2: hi(W_last), rows 8-10: lo(W_last)], slots (s, 1), (s, 2) zero, b1 = conv_hr's bias, b2 = [b_last, 0 ...]
 * (refvsr_amd/packing.py:pack_conv_hr_last); base_lr / out as in refvsr_conv_last. */
int refvsr_conv_hr_last(const void* src, int h, int w, const void* blob, float act_slope, const float* base_lr, int bh, int bw,
                        float* out, void* stream);
/* Result formats of the output head (ABI 14; extension -- the reference returns fp32 and its consumers quantise on the CPU:
 * evaluation/eval_qual_quan.py:117-119 hands cv2.imwrite `output * 255`, i.e. saturate_cast<uchar> = round to nearest even).
 * The _fmt forms of the two head launches store the planar [3][h][w] result as fp32 (== the plain forms), fp16, or uint8 =
 * rint(255 v) of the clamped fp32 value -- the bytes the reference's PNG writer produces -- so that 1/2 or 1/4 of the bytes cross
 * PCIe; refvsr_convert_result does the same conversion on an fp32 result (the generic head of configurations without a fused one). */
enum { REFVSR_RESULT_F32 = 0, REFVSR_RESULT_F16 = 1, REFVSR_RESULT_U8 = 2 };
int refvsr_conv_last_fmt(const void* src, int c, int h, int w, const void* blob, const float* base_lr, int bh, int bw,
                         void* out, int out_fmt, void* stream);
int refvsr_conv_hr_last_fmt(const void* src, int h, int w, const void* blob, float act_slope, const float* base_lr, int bh, int bw,
                            void* out, int out_fmt, void* stream);
int refvsr_convert_result(const float* src, size_t n, int out_fmt, void* out, void* stream);

/* The confidence fusions in ONE launch (ABI 10): conf_fusion / conf_fusion2 / conf_fusion_BWFW of RefVSR.py:47-52, called at
 * :130, :141-142 and :107-109 as  conv_{16->C}(lrelu(conv_{2->16}(cat[conf_a, conf_b])))  on the LR grid (up = 1) and on
 * clamp(F.interpolate(cat[conf_a, conf_b], scale_factor=2, mode='bicubic'), 0, 1) (up = 2).  conf_a / conf_b: planar fp32
 * [h][w]; w0 / b0: the 2 -> 16 conv's fp32 weights [16][2][3][3] / bias [16] (device); blob: the 16 -> cout blob of refvsr_conv24 /
 * refvsr_conv48 (cout = 24 | 48); alpha: fp16 HWC [up h][up w][cout]; conf_max (optional, up = 1 only): max(conf_a, conf_b)
 * [h][w], the propagated confidence of RefVSR.py:147.  The 16-channel map, the concatenated and the up-sampled pair never
 * exist in HBM; results are bit-identical to torch.cat + [refvsr_resize +] refvsr_conv_direct_f32 + refvsr_conv24/48
 * [+ refvsr_max2] (tests/test_gpu_ops.py). */
int refvsr_conf_alpha(const float* conf_a, const float* conf_b, int h, int w, int up, const float* w0, const float* b0,
                      float slope0, const void* blob, int cout, float slope1, void* alpha, float* conf_max, void* stream);
/* refvsr_conf_alpha over `batch` map pairs (ABI 11, cout = 24): conf_a / conf_b / alpha / conf_max (NULL array = none) are host
 * arrays of `batch` device pointers. */
int refvsr_conf_alpha_batch(const float* const* conf_a, const float* const* conf_b, int batch, int h, int w, int up, const float* w0,
                            const float* b0, float slope0, const void* blob, int cout, float slope1, void* const* alpha,
                            float* const* conf_max, void* stream);
/* The same for 48 output channels (mid_channels = 48: configs/config_RefVSR_{L1,MFID,MFID_8K}.py -- the two convs of every
 * ResidualBlockNoBN, sr_backbone_utils.py:42-97, and conf_fusion*.1): (c0, c1) in {(48,0), (16,0)}; out / mul / res are
 * 48-channel maps; blob: [S x 6 fragments x 64 lanes x 8 halfs][64 bias floats], fragments [hi | lo] of output channels 0-15,
 * 16-31, 32-47, K-blocks by refvsr_conv24_kblock(ncg, s, q).  48 -> 48 keeps its 84 KB weight set resident next to a
 * 16 x 32-pixel tile walked by sixteen waves (one workgroup per CU).
 * Two-source forms (ABI 10): (8, 48) -- the input conv of ResidualBlocksWithInputConv on cat([lr, feat]), RefVSR.py:340-343: the
 * same blob layout on the ncg = 7 plan (18 K-steps, one zero block per tap) -- and (48, 48) -- feat_fusion*.0 / feat_fusion2_1 /
 * fusion_UP on cat([a, b]), RefVSR.py:53-62,87: one conv's 166 KB of weights have no resident form, so the OUTPUT channels are
 * computed in two halves of 24 on blockIdx.y; blob = two blobs of the 24-output layout (refvsr_conv24's: [27 K-steps x 3 fragments
 * x 64 lanes x 8 halfs][32 bias floats], ncg = 12 plan) for channels 0-23 and 24-47, back to back. */
/* The same for 32 output channels (AlignedConv2d, RefVSR_/alignment.py:18-24,53-100: the RGB stem and the 32 -> 32 convs of its
 * ResBlocks, at the 2x / HD resolutions): (c0, c1) in {(32,0), (8,0)}; blob: [S x 4 fragments x 64 lanes x 8 halfs][32 bias floats],
 * fragments [hi | lo] of output channels 0-15, 16-31. */
int refvsr_conv32_supported(int c0, int c1);
int refvsr_conv32_blob_bytes(int c0, int c1);
int refvsr_conv32(const void* src0, int c0, const void* src1, int c1, int h, int w, const void* blob, float act_slope,
                  const void* mul, const void* res, float post_slope, void* out, void* stream);
int refvsr_conv48_supported(int c0, int c1);
int refvsr_conv48_blob_bytes(int c0, int c1);
int refvsr_conv48(const void* src0, int c0, const void* src1, int c1, int h, int w, const void* blob, float act_slope,
                  const void* mul, const void* res, float post_slope, void* out, void* stream);
/* PixelShufflePack (mmedit upsample.py:36-51; RefVSR.py:89-90 upsample1 / upsample2, used at :114,116,138): the C -> 4 C 3x3 conv
 * and F.pixel_shuffle(2) in one launch of the same kernel family, C = 24 | 48: src [h][w][C] fp16 HWC -> out [2h][2w][C], bias +
 * optional leaky activation (act_slope in [0, 1], 1 = none: it commutes with the shuffle, RefVSR.py:116).  blobs: refvsr_conv_shuffle2_blob_bytes(C) bytes = 2 (C = 24) or 4 (C = 48) conv48-style blobs of 48 output rows each,
 * back to back, rows ordered sub-pixel-major: C = 24: blob z = output row parity dy, rows [dx = 0: channels 0-23][dx = 1: 0-23]
 * (row R of blob z = weight row 4 (R % 24) + 2 z + R / 24 of the reference conv); C = 48: blob z = sub-pixel 2 dy + dx, row R =
 * weight row 4 R + z (refvsr_amd/packing.py:pack_conv_shuffle2).  Same arithmetic as refvsr_conv_mfma's SHUFFLE2 output mode
 * up to fp32 summation order. */
int refvsr_conv_shuffle2_supported(int c);
int refvsr_conv_shuffle2_blob_bytes(int c);
int refvsr_conv_shuffle2(const void* src, int c, int h, int w, const void* blobs, float act_slope, void* out, void* stream);
/* refvsr_conv_shuffle2 over `batch` maps (ABI 11, c = 24): src / out host arrays of `batch` device pointers. */
int refvsr_conv_shuffle2_batch(const void* const* src, int batch, int c, int h, int w, const void* blobs, float act_slope,
                               void* const* out, void* stream);
/* Debug knob (no reference counterpart): when buf != NULL every workgroup of the PROBE instantiations of the fused-block kernels records s_memtime
 * stamps (entry, loads issued, loads landed, conv1 K loop, conv1 epilogue, barrier, conv2 K loop, stores issued) of its
 * iter-th tile at buf[12 * workgroup + i] (uint64; [8], [9] = 100 MHz s_memrealtime at entry / exit, [10] = s_memtime at exit) -- tools/probe_resblock.py.  NULL switches it off (default). */
int refvsr_set_probe(void* buf, int iter);

/* fp32 direct convolution on planar maps (VGG feature extractor + MeanShift of FeatureMatching,
 * attention.py:28-50,62-70, and the 2->16 confidence convs RefVSR.py:47-52).  Kept in fp32 because
 * the arg-max of the matching is discontinuous in these features.
 * w: fp32 [cout][cin][k][k]; out: planar fp32 [cout][ho][wo] or nhwc16 [ho][wo][out_c]. */
int refvsr_conv_direct_f32(const float* src, int cin, int h, int w,
                           const float* wgt, const float* bias, int cout, int ksize, int stride, int pad,
                           float act_slope, void* out, int out_nhwc16, int out_c, void* stream);
/* 1x1 conv cin -> 16 + LeakyReLU in fp32 (ABI 10): the map64 / map128 block that ends the matching's feature extractor
 * (RefVSR_/attention.py:41-42).  src: fp32 HWC [h][w][cin] (cin % 4 == 0, the f32 conv mode's output), wgt: fp32 [16][cin], bias [16];
 * out: planar fp32 [16][h][w].  Channels are summed in order with fp32 FMAs. */
int refvsr_conv1x1_f32(const float* src, int cin, int h, int w, const float* wgt, const float* bias, float act_slope,
                       float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Layout conversion
 * ------------------------------------------------------------------------------------------ */
/* planar fp32 [c][h][w] -> nhwc16 [h][w][cs] (channels >= c zero-filled). */
int refvsr_pack_nhwc16(const float* src, int c, int h, int w, void* dst, int cs, void* stream);
/* planar fp32 [c][h][w] -> fp32 HWC [h][w][cs], cs % 4 == 0 (input of the f32 conv mode). */
int refvsr_pack_nhwc32(const float* src, int c, int h, int w, float* dst, int cs, void* stream);
/* nhwc16 [h][w][cs] -> planar fp32 [c][h][w]. */
int refvsr_unpack_nhwc16(const void* src, int h, int w, int cs, int c, float* dst, void* stream);

/* ------------------------------------------------------------------------------------------
 * Resampling (F.interpolate / avg_pool2d / max_pool2d call sites)
 * ------------------------------------------------------------------------------------------ */
enum { REFVSR_RS_BICUBIC = 0, REFVSR_RS_BILINEAR = 1, REFVSR_RS_BILINEAR_AC = 2, REFVSR_RS_NEAREST = 3 };
/* dst = post(interp(src)); post: (v - mean[c]) / std[c] if mean != NULL (host arrays of c floats),
 * then v * mul, then clamp to [0,1] if clamp01.  src_scale_{y,x}: source step per output sample
 * (1/scale_factor, or in/out when the reference passes size=); ignored for BILINEAR_AC.
 * dst is planar fp32 [c][oh][ow], or nhwc16 [oh][ow][out_c] when out_nhwc16 != 0. */
int refvsr_resize(const float* src, int c, int h, int w, void* dst, int oh, int ow, int mode,
                  float src_scale_y, float src_scale_x, const float* mean, const float* std,
                  const float* chan_mul, int clamp01, int out_nhwc16, int out_c, void* stream);
int refvsr_avgpool2(const float* src, int c, int h, int w, float* dst, void* stream);
int refvsr_maxpool2(const float* src, int c, int h, int w, float* dst, void* stream);
/* flags[i] &= (a[i][0..n_bytes) == b[i][0..n_bytes)) for i < n_pairs (<= 32) in one launch; the caller presets
 * flags to 1.  a, b: HOST arrays of device pointers (16-byte aligned buffers, n_bytes % 16 == 0).
 * Keys the per-frame cache (frames of consecutive sliding windows are recognised by content). */
int refvsr_buffers_equal(const void* const* a, const void* const* b, int n_pairs, size_t n_bytes,
                         int32_t* flags, void* stream);
/* out = max(a, b) elementwise on n floats (confidence accumulation, RefVSR.py:147). */
int refvsr_max2(const float* a, const float* b, float* out, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Inter-frame alignment
 * ------------------------------------------------------------------------------------------ */
/* models/utils.py:35-43 `warp`: linspace(-1,1) base grid + flow/((Win-1)/2), grid_sample(bilinear,
 * zeros, align_corners=False).  Output size = flow size [hf][wf]; input may be a different size
 * (RefVSR.py:254).  flow: planar fp32 [2][hf][wf]. */
int refvsr_warp_nhwc16(const void* x, int hin, int win, int cs, const float* flow, int hf, int wf,
                       void* out, void* stream);
/* warp(x, F.interpolate(flow_lr, scale_factor=2, mode='bilinear', align_corners=True) * 2) in one launch (ABI 10): the 2x
 * propagation state is warped by the up-sampled LR flow (RefVSR.py:220,254,259); the [2][2hl][2wl] flow map is evaluated per
 * pixel instead of being written and read back.  flow_lr: planar fp32 [2][hl][wl]; out [2hl][2wl][cs].  Bit-identical to
 * refvsr_resize(BILINEAR_AC, chan_mul 2) + refvsr_warp_nhwc16. */
int refvsr_warp_nhwc16_up2(const void* x, int hin, int win, int cs, const float* flow_lr, int hl, int wl,
                           void* out, void* stream);
int refvsr_warp_planar(const float* x, int c, int hin, int win, const float* flow, int hf, int wf,
                       float* out, void* stream);
/* The three warps over `batch` (map, flow) pairs in one launch each (ABI 11; blockIdx.z = pair): x / flow / out are host arrays of
 * `batch` device pointers, all pairs of one geometry.  Pair b == the single call, bit for bit. */
int refvsr_warp_nhwc16_batch(const void* const* x, int batch, int hin, int win, int cs, const float* const* flow, int hf, int wf,
                             void* const* out, void* stream);
int refvsr_warp_nhwc16_up2_batch(const void* const* x, int batch, int hin, int win, int cs, const float* const* flow_lr, int hl, int wl,
                                 void* const* out, void* stream);
int refvsr_warp_planar_batch(const float* const* x, int batch, int c, int hin, int win, const float* const* flow, int hf, int wf,
                             float* const* out, void* stream);
/* One SPyNet pyramid level input (SPyNet.py:83-103): flow_up = 2*bilinear_x2(flow_prev, align_corners)
 * (or zeros when flow_prev == NULL), warped = flow_warp(supp, flow_up, border, align_corners=True)
 * (mmedit flow_warp.py:6-47); writes cat[ref, warped, flow_up] as nhwc16 [h][w][8] and flow_up as
 * planar fp32 [2][h][w].  ref/supp: planar fp32 [3][h][w]; flow_prev: planar [2][h/2][w/2]. */
int refvsr_spynet_level_input(const float* ref, const float* supp, const float* flow_prev, int h, int w,
                              void* out8, float* flow_up, void* stream);
/* The same for `batch` (1..8; 4 until ABI 10) independent (ref, supp) pairs of one pyramid level in one launch: ref / supp are host arrays of
 * `batch` device pointers; flow_prev [batch][2][h/2][w/2] (or NULL), out8 [batch][h][w][8], flow_up [batch][2][h][w] are
 * contiguous batches.  Image b == refvsr_spynet_level_input of pair b, bit for bit. */
int refvsr_spynet_level_input_batch(const float* const* ref, const float* const* supp, int batch, const float* flow_prev,
                                    int h, int w, void* out8, float* flow_up, void* stream);

/* ------------------------------------------------------------------------------------------
 * Reference matching  (FeatureMatching.forward, RefVSR_/attention.py:72-91)
 * ------------------------------------------------------------------------------------------ */
#define REFVSR_MATCH_KP 152       /* 144 = 16 ch x 3x3 patch, padded to 152 halfs (304-byte rows)   */
#define REFVSR_MATCH_ROWCHUNK 256 /* reference rows are padded to a multiple of this               */
#define REFVSR_MATCH_COLBLOCK 512 /* LR columns are padded to a multiple of this                    */
/* feat: planar fp32 [16][h][w].  Writes rows [h*w][KP] fp16 of L2-normalised reflect-padded 3x3
 * patches (channel order c*9+ky*3+kx, RefVSR_/utils.py:29-57), inv_norm[h*w] = 1/max(|p|,1e-12) and, when rows_lo != NULL,
 * the low halves of the fp16 hi + lo split of the same normalised patches, rows_lo [h*w][KP] fp16 =
 * fp16((p - rows) * 2^11) (second operand of refvsr_match_exact; allocate it padded like rows). */
int refvsr_match_patches(const float* feat, int h, int w, void* rows, float* inv_norm, void* rows_lo, void* stream);
/* Fused cosine GEMM + column top-2 (never materialises the [n_ref x n_lr] matrix).
 * ref_rows: [n_ref_pad][KP], lr_rows: [n_lr_pad][KP] (pads zero).  row_splits >= 1 partitions the
 * reference rows over blockIdx.y.  cand_idx / cand_val: [n_lr][2*row_splits] (first-max-wins order). */
int refvsr_match_top2(const void* ref_rows, int n_ref, const void* lr_rows, int n_lr, int row_splits,
                      int32_t* cand_idx, float* cand_val, void* stream);
/* Exact fp32 re-rank of the candidates: conf[p] = max_c <lr_patch p, ref_patch c> (normalised),
 * idx[p] = that candidate (smallest index on ties, like torch.max).  When flagged != NULL (int32 [1 + h*w], [0] = 0 on
 * entry): columns whose re-ranked maximum does not clear the runner-up's fp16 score (cand_val) by `margin` -- i.e. where
 * a row outside the candidate list could still be the true arg-max -- are appended to flagged[1..], flagged[0] counts. */
int refvsr_match_refine(const float* lr_feat, int h, int w, const float* ref_feat, int hr, int wr,
                        const float* inv_lr, const float* inv_ref, const int32_t* cand_idx, const float* cand_val,
                        int ncand, float margin, int32_t* flagged, float* conf, int32_t* idx, void* stream);
/* Exhaustive fp32-grade arg-max for the flagged columns: both operands split into fp16 hi + lo, three fp16 MFMAs with fp32
 * accumulation per product (dropped term 2^-22).  The winner of every flagged column is re-evaluated with the exact fp32
 * dot product of refvsr_match_refine and replaces that column's conf / idx if it is better (smaller index on ties).
 * lr_rows / lr_rows_lo, ref_rows / ref_rows_lo: the fp16 hi / lo patch rows of refvsr_match_patches (reference rows padded
 * to a multiple of 256); keys: uint64 [h*w] zeroed scratch.  The flagged count is read on the device (no host
 * synchronisation); with no flagged column the launches are no-ops. */
int refvsr_match_exact(const float* lr_feat, int h, int w, const float* ref_feat, int hr, int wr, const void* lr_rows,
                       const void* lr_rows_lo, const void* ref_rows, const void* ref_rows_lo, const float* inv_lr,
                       const float* inv_ref, const int32_t* flagged, void* keys, float* conf, int32_t* idx, void* stream);
/* Unfused fp32 reference kernel of the same op (test / debugging aid, O(n_ref*n_lr*144) VALU). */
int refvsr_match_naive(const float* lr_feat, int h, int w, const float* ref_feat, int hr, int wr,
                       float* conf, int32_t* idx, void* stream);

/* ------------------------------------------------------------------------------------------
 * Reference alignment  (AlignedAttention / AlignedConv2d, attention.py:131-159, alignment.py:39-178)
 * ------------------------------------------------------------------------------------------ */
/* unfold(k=s,stride=s) -> gather(index) -> fold == block gather:
 * out[s*y+ky][s*x+kx][:] = value[s*ry+ky][s*rx+kx][:], (ry,rx) = divmod(idx[y*gw+x], wv/s). */
int refvsr_block_gather_nhwc16(const void* value, int hv, int wv, int cs, const int32_t* idx, int gh, int gw,
                               int s, void* out, void* stream);
/* same gather on a planar fp32 RGB frame, written as nhwc16 [gh*s][gw*s][8] (3 valid channels; out8, may be NULL)
 * and / or as an exact planar fp32 copy [3][gh*s][gw*s] (out_planar, may be NULL: the `vis` samples RefVSR.py:305-309). */
int refvsr_block_gather_rgb(const float* value, int hv, int wv, const int32_t* idx, int gh, int gw, int s,
                            void* out8, float* out_planar, void* stream);
/* affine-deformable bilinear patch sampler (alignment.py:53-100,102-178).
 * x: nhwc16 [h*ks][w*ks][cs]; affine: planar fp32 [3][h][w] already (+1, clamped to [-3,3]). */
int refvsr_aligned_sample(const void* x, int h, int w, int ks, int cs, const float* affine, void* out,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * RefVSR_IR: EDVR-M feature extractor pieces that are not convolutions (models/archs/edvr_net.py)
 * ------------------------------------------------------------------------------------------ */
/* Sampling half of mmcv's modulated deformable convolution (edvr_net.py:49-56, ModulatedDCNPack; 3x3, stride 1, pad 1):
 * cols[y][x][k*c + ch] = sigmoid(mask) * bilinear(x, tap position + offset) for tap k, 8 channels per deformable group;
 * offset_mask = raw conv_offset output, planar fp32 [3*groups*9][h][w] ([o1 | o2 | mask]).  The contraction with the
 * [cout][c*9] weight is a 1x1 refvsr_conv_mfma over cols. */
int refvsr_dcn_sample(const void* x, int h, int w, int c, const float* offset_mask, int deform_groups, void* cols,
                      void* stream);
/* TSAFusion temporal attention (edvr_net.py:259-272): out[.., i*c + ch] = aligned_i * sigmoid(sum_ch emb_i * emb_ref);
 * aligned / emb: HOST arrays of t device pointers (nhwc16 [npix][c]). */
int refvsr_tsa_weight(const void* const* aligned, const void* const* emb, const void* emb_ref, int t, int c, int npix,
                      void* out, void* stream);
/* MaxPool2d / AvgPool2d(3, stride 2, padding 1) of an nhwc16 map into channels [c_off, c_off + c) of out (channel
 * stride out_c) (edvr_net.py:214-215,277-279). */
int refvsr_pool3s2_nhwc16(const void* x, int h, int w, int c, void* out, int out_c, int c_off, int is_max, void* stream);
/* nn.Upsample(x2, bilinear, align_corners=False) * mul of an nhwc16 map (edvr_net.py:131,179-181,246). */
int refvsr_up2_bilinear_nhwc16(const void* x, int h, int w, int c, float mul, void* out, void* stream);
/* out = feat * sigmoid(attn) * 2 + add on n halfs (edvr_net.py:294-299). */
int refvsr_tsa_blend(const void* feat, const void* attn, const void* add, size_t n, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REFVSR_HIP_H */
