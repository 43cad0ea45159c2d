/*
 * dsg.h -- C ABI of libdsg.so, the MI355X (gfx950) denoising engine for DriveSceneGen's
 * scene-raster U-Net.
 *
 * The reference (SS47816/DriveSceneGen) has no FFI of its own: its hot path calls Python objects
 * from diffusers 0.20.0.  Each entry point below names the reference call site whose arithmetic it
 * replaces (paths relative to the reference checkout) -- the Python shim in drivescenegen_amd/
 * binds these with ctypes.
 *
 * Conventions
 *  - every tensor is a raw DEVICE pointer, fp32, contiguous; activations NCHW, as the reference's
 *    checkpoints and call sites use (train.py:39-57, training_pipeline.py:84).  The mixed-precision modes
 *    (dsg_dtype: the reference trains under Accelerator(mixed_precision='fp16'), train.py:24 /
 *    training_pipeline.py:48-49; BASELINE.json configs[4] names bf16) keep that boundary: parameters,
 *    [N,C,H,W] tensors, GroupNorm statistics, softmax, accumulators and the optimizer stay fp32, only the
 *    engine's channel-blocked intermediates and the matrix-core operands are 16-bit (torch.autocast's split);
 *  - the caller owns every buffer, including workspaces; nothing is allocated per call; only
 *    dsg_unet_* handles own device memory (their re-laid-out weights) until dsg_unet_destroy;
 *  - `stream` is a hipStream_t passed as void*; calls enqueue and return without synchronising;
 *  - every function returns DSG_OK (0) or a negative dsg_status; dsg_last_error() gives the
 *    thread-local message.  No exception or abort crosses this boundary.
 */
#ifndef DSG_H_
#define DSG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  DSG_OK = 0,
  DSG_ERR_INVALID_ARG = -1,
  DSG_ERR_UNSUPPORTED_SHAPE = -2,
  DSG_ERR_WORKSPACE_TOO_SMALL = -3,
  DSG_ERR_HIP = -4,
  DSG_ERR_NOT_READY = -5
} dsg_status;

/* Arithmetic type of the matrix-core products (and storage type of channel-blocked tensors):
 *   DSG_F32   fp32-equivalent: fp32 tensors; products as an fp16x2 split on the f16 matrix cores, or the f32 MFMA
 *   DSG_BF16  bf16 operands / tensors, fp32 accumulate             (BASELINE.json configs[4])
 *   DSG_F16   fp16 operands / tensors, fp32 accumulate             (train.py:24 mixed_precision='fp16') */
typedef enum { DSG_F32 = 0, DSG_BF16 = 1, DSG_F16 = 2 } dsg_dtype;

int dsg_version(void);
const char* dsg_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Fused convolution: the Conv2d call sites of UNet2DModel.forward (train.py:39-57 builds it,
 * training_pipeline.py:84 calls it): conv_in / resnet conv1, conv2 / conv_shortcut /
 * downsamplers.0.conv (stride 2) / upsamplers.0.conv (after nearest x2) / attention q,k,v,out
 * projections (1x1 over [N,C,H*W]) / conv_out.
 *
 *   dst = conv(act(affine(cat(src0, src1)[optionally nearest-upsampled x2])), W) + bias
 *         (+ temb[n, cout]) (+ residual)
 *
 * affine/act is the GroupNorm-apply(+SiLU) of the preceding norm folded into the gather
 * (gn_scale_shift[n][cin] = {rstd*gamma, beta - mean*rstd*gamma}); zero padding is applied after it.
 * Weight is in the engine layout produced by dsg_conv_weight_relayout: [Cin][k*k][Cout].
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const float* src0;            /* [N, c0, hin, win] */
  const float* src1;            /* optional [N, c1, hin, win] concatenated after src0 on channels */
  int32_t c0, c1;
  int32_t n, hin, win;          /* source dims (before the optional upsample) */
  int32_t upsample;             /* 1: nearest x2 of the source before the conv (Upsample2D);
                                   2: zero-stuffed x2 (source at even positions) -- the data gradient of a
                                      stride-2 conv is a stride-1 conv over it */
  int32_t ksize;                /* 3 or 1; padding = ksize/2 */
  int32_t stride;               /* 1 or 2 (Downsample2D) */
  int32_t cout;
  const float* weight;          /* [c0+c1][ksize*ksize][weight_cout_stride]; may be NULL when one of the weight_h2* operand
                                   images below serves the call (otherwise DSG_ERR_INVALID_ARG, before anything is
                                   launched): a training step need not re-lay-out weights no kernel reads */
  int32_t weight_cout_stride;   /* 0 = cout; >= cout when the matrix is zero-padded (conv_out: 4 -> 32) */
  const float* bias;            /* [cout] or NULL */
  const float* gn_scale_shift;  /* optional [N][c0+c1][2] */
  int32_t silu;                 /* 1: SiLU after the affine */
  const float* temb;            /* optional: temb[n*temb_stride + cout_index] added per (n, cout) */
  int32_t temb_stride;
  const float* residual;        /* optional [N, cout, hout, wout] */
  float* dst;                   /* [N, cout, hout, wout] ([N, cout, hout/2, wout/2] with pool2) */
  int32_t pool2;                /* 1: 2x2 sum-pool of the result (adjoint of the nearest x2 upsample) */
  const void* weight_h2;        /* optional: the same weights pre-split for the fp16x2 matrix-core path
                                   (dsg_conv_weight_relayout_h2); used for stride-1 3x3 / 1x1 convs with cin % 16 == 0, cout % 64 == 0 */
  int32_t weight_h2_cout_stride; /* 0 = cout padded to 64; else the row length (in couts, a multiple of 64) of the wider
                                   pre-split matrix that weight_h2 points into (a column window: weight_h2 =
                                   matrix + 8 * first_column halfs; first_column + cout rounded up to 64 must not
                                   exceed the row length -- whole 64-column tiles are read) */
  const void* weight_h2_fold;   /* optional, upsample == 1 only: the 3x3 weights folded into four 2x2 phase kernels
                                   (dsg_conv_weight_relayout_h2_fold); the conv then runs on the low-resolution
                                   grid with 16 instead of 36 tap products per input pixel */
  double* stats_out;            /* optional [N][cout][tiles][2]: per-tile (sum, sum of squares) of dst, so that the
                                   GroupNorm that follows needs no pass of its own over dst (dsg_gn_finalize_parts).
                                   Only where dsg_conv2d_stats_tiles reports tiles > 0. */
  int32_t src_layout;           /* layout of src0 / src1: 0 = [N, C, H, W]; 1 = channel-blocked [N, C/8, H, W, 8] (C % 8
                                   == 0): the 8 channels of a pixel are 32 contiguous bytes, so a conv's halo-patch
                                   gather is two 16-byte loads per pixel instead of eight 4-byte ones (measured 5.4x
                                   on the load side, tools/membench.hip).  dsg_unet_forward keeps its intermediate
                                   activations in this layout; the public tensors stay [N, C, H, W]. */
  int32_t dst_layout;           /* the same for dst and residual */
  const void* weight_h2_s2;     /* optional, stride == 2 with both src and dst channel-blocked, or (fp32) both [N, C, H, W]: the 3x3 weights laid out for
                                   the 2x2 conv over the space-to-depth image (dsg_conv_weight_relayout_h2_s2); the
                                   down-sampler conv then runs on the split path too */
  int32_t compute_dtype;        /* dsg_dtype.  DSG_F32 (0): everything above as described.  DSG_BF16 / DSG_F16: the
                                   channel-blocked tensors (src_layout / dst_layout == 1: src0, src1 / dst, residual) are
                                   stored in that 16-bit type, [N, C, H, W] tensors stay fp32; products run once on the
                                   bf16 / f16 matrix cores with fp32 accumulation; GroupNorm affine + SiLU are evaluated in
                                   fp32 before the operand is rounded; bias / temb / scale_shift / stats stay fp32 / fp64.
                                   weight_h2* must then come from dsg_conv_weight_pack(..., the same dtype).  Served:
                                   the matrix-core shapes with at least one channel-blocked side, conv_in (fp32 image ->
                                   blocked) and conv_out (blocked -> fp32 image); anything else is DSG_ERR_UNSUPPORTED_SHAPE. */
  /* Range guard of the fp32-equivalent split path (compute_dtype == DSG_F32), all optional:
     src_bound / src_bound1  [N] per-image upper bounds of max|src0| / max|src1| as IEEE-754 bits of a non-negative
                             float.  Used by calls WITHOUT gn_scale_shift (their source is not normalised): when the bound
                             leaves [2^-6, 2^12] the patch is scaled by the power of two that brings it to ~1 before the
                             fp16 split and the result scaled back -- exact, so large or tiny activations neither
                             overflow the fp16 pieces nor lose precision.  NULL: no guard (|x| < 65504 is then required).
                             Bounds come from the statistics a GroupNorm needs anyway: dsg_gn_finalize_parts_bound,
                             dsg_range_bound_from_stats. */
  const uint32_t* src_bound;
  const uint32_t* src_bound1;
  /* Split-K for grids smaller than the chip (the reference's own sampling calls run at batch 1 and 5): optional scratch
     of dsg_conv2d_splitk_bytes(args) bytes.  With it, a call on channel-blocked fp32 tensors whose tile grid covers at
     most half the CUs contracts its K-chunks in 2-4 parallel slices (fp32 partial sums in the scratch) and a reduce
     pass adds them in slice order together with bias / temb / residual and writes stats_out.  The accumulation order
     then differs from the one-slice kernel's (round-off class, <= 1e-6 relative); results still do not depend on
     anything but the call's own shape.  NULL: never split. */
  void* splitk_ws;
  size_t splitk_ws_bytes;
  /* Fused resnet shortcut (optional; diffusers ResnetBlock2D with in != out channels, i.e. the first resnet of a wider
     down block and every up-block resnet of train.py:39-57's network: output = conv_shortcut(input) + conv2(act(norm2(h)))):
       dst = conv3x3(act(affine(src))) + bias (+ temb) + conv1x1(cat(sc_src0, sc_src1), sc_weight_h2) + sc_bias
     The 1x1 over the resnet's UN-normalised input is contracted in the same kernel, on the same accumulators: the
     shortcut's result is never written to HBM nor read back as `residual` (which must be NULL).  Served for the calls
     dsg_conv2d_fuses_shortcut accepts (a resnet's conv2 on channel-blocked tensors); anything else with sc_weight_h2
     set is DSG_ERR_UNSUPPORTED_SHAPE.  sc_weight_h2 = dsg_conv_weight_pack(kind 0, ksize 1, compute_dtype) of the
     [cout][sc_c0 + sc_c1] weight; sc_src* are [N, sc_c*, hout, wout] in src_layout; sc_src_bound* as src_bound* above. */
  const void* sc_src0;
  const void* sc_src1;
  int32_t sc_c0, sc_c1;
  const void* sc_weight_h2;
  const float* sc_bias;
  const uint32_t* sc_src_bound;
  const uint32_t* sc_src_bound1;
  /* Pre-staged operand image (optional; fp32-equivalent mode, channel-blocked tensors): the output of
     dsg_conv_operand_prepare for this call's sources -- GroupNorm affine + SiLU (or the range guard's pre-scale) and the
     fp16x2 split, which the kernel otherwise performs in its staging pass ONCE PER COUT TILE, done once.  The kernel then
     DMAs its halo patches from the image straight into LDS and stages nothing.  src0 / src1 / gn_scale_shift / silu /
     src_bound* still describe the call (and must be what the image was prepared from); results are bit-identical to the
     same call without the image.  Served for the calls dsg_conv2d_takes_operand accepts (a resnet's conv1 / conv2 and the
     folded up-sampler conv of the levels with >= 256 output channels at batch sizes that fill the chip); anything else
     with src_operand set is DSG_ERR_UNSUPPORTED_SHAPE. */
  const void* src_operand;
  /* GroupNorm-backward statistics from the epilogue (optional; the DATA-GRADIENT call of a conv whose forward read
     a = silu(GroupNorm(x)): training_pipeline.py:86 `accelerator.backward(loss)` through ResnetBlock2D's norm1 / norm2).
     dst of such a call is dA, the gradient w.r.t. the activated tensor, and the norm's backward starts with a pass of its own
     over x and dA for per-(n, c) sums of du = dA * silu'(x * scale + shift) and du * xhat.  With gnb_x0 set (gnb_x1 / gnb_c0:
     the second tensor of a concatenated x and the channel count of the first) the kernel reads the x of its output tile
     (same layout / dtype as dst) and gnb_ss ([N][cout][2] scale, shift; gnb_silu: 0 = affine only) in its epilogue and
     writes per-tile partials (sum du, sum du * x) to stats_out ([N][cout][tiles][2], tiles = dsg_conv2d_stats_tiles) -- the
     RAW second moment: dsg_gn_bwd*_parts applies xhat = (x - mean) * rstd to the sums.  dst is unchanged.  Served for the
     calls dsg_conv2d_gnb_supported accepts; anything else with gnb_x0 set is DSG_ERR_UNSUPPORTED_SHAPE. */
  const void* gnb_x0;
  const void* gnb_x1;
  int32_t gnb_c0;
  const float* gnb_ss;
  int32_t gnb_silu;
  /* With stride == 2 and weight_h2_s2: 0 = the 3x3 pad-1 conv above.  1 = the call is the DATA GRADIENT of an up-sampler
     (nearest-2x followed by a 3x3 conv; reference: diffusers Upsample2D as built by DriveSceneGen/utils/model/unet_2d.py's
     up blocks): src0 is dY at full resolution [N, c0, 2h, 2w], dst is dX [N, cout, h, w] (c0 = the up-sampler conv's cout,
     cout = its cin), weight_h2_s2 is dsg_conv_weight_pack kind 5 of the conv's weight: one 4x4 stride-2 window per low-resolution
     pixel, 16 taps instead of the 36 of "3x3 data gradient at full resolution, then sum the 2x2 pixels".  Channel-blocked
     tensors only (`weight` is not read); DSG_ERR_UNSUPPORTED_SHAPE where the space-to-depth kernel does not take the shape. */
  int32_t s2_window4;
} dsg_conv_args;

int dsg_conv2d_fwd(const dsg_conv_args* a, void* stream);
/* [N, C, H, W] <-> [N, C/8, H, W, 8] (to_blocked: 1 | 0); C % 8 == 0; src != dst */
int dsg_layout_convert(const float* src, float* dst, int32_t n, int32_t c, int32_t hw, int32_t to_blocked, void* stream);
/* the same with the channel-blocked side stored as blocked_dtype (the [N, C, H, W] side is always fp32) */
int dsg_layout_convert_dt(const void* src, void* dst, int32_t n, int32_t c, int32_t hw, int32_t to_blocked,
                          int32_t blocked_dtype, void* stream);
/* Number of spatial tiles per (n, cout) this call would write into stats_out; 0 when the kernel that serves the
 * call cannot produce the statistics (then run dsg_gn_channel_stats on dst instead). Host-only.  (Set splitk_ws
 * before asking: the split-K path has its own tile count.) */
int dsg_conv2d_stats_tiles(const dsg_conv_args* a, int32_t* tiles);
/* Scratch bytes the call would use for split-K (0: it would not split). Host-only; independent of splitk_ws. */
int dsg_conv2d_splitk_bytes(const dsg_conv_args* a, size_t* bytes);
/* *yes = 1 when a call with these arguments (sc_src0 / sc_c0 / sc_c1 / sc_weight_h2 filled in) takes the fused-shortcut
 * kernel, 0 when the shortcut has to run as a 1x1 call of its own with its result passed as `residual`.  Host-only; the
 * answer depends on shapes, layouts, dtype and the split-K decision of the call.  That decision is taken from the grid
 * size, i.e. from the batch: with splitk_ws set, whether a resnet's shortcut is fused (a different summation order, equal
 * to fp32 round-off) can change with the batch size; without splitk_ws (DSG_UNET_BATCH_INVARIANT plans) it cannot. */
int dsg_conv2d_fuses_shortcut(const dsg_conv_args* a, int32_t* yes);
/* *yes = 1 when a call with these arguments is served by a kernel that reads a pre-staged operand image AND staging once
 * pays (the patch would otherwise be staged by >= 4 workgroups); the caller then prepares the image and sets src_operand.
 * Host-only; set splitk_ws before asking.  Reference: the resnet convs of train.py:39-57's network, run at
 * training_pipeline.py:84 / inside DDPMPipeline.__call__ (training_pipeline.py:26-32, generation.py:14-20). */
int dsg_conv2d_takes_operand(const dsg_conv_args* a, int32_t* yes);
/* *yes = 1 when a call with these arguments (gnb_* filled in, stats_out not needed yet) is served by a kernel with the
 * GroupNorm-backward epilogue: a stride-1 3x3 conv without a norm in front, fused shortcut, operand image or split-K;
 * 16-bit modes: every tensor channel-blocked; fp32 mode: every tensor [N, C, H, W]; maps a multiple of 32 columns wide; and
 * gnb_c0 a multiple of 32 (the epilogue reads x in 32-channel slabs: gnb_x0 and gnb_x1 may meet inside a channel tile).  Host-only. */
int dsg_conv2d_gnb_supported(const dsg_conv_args* a, int32_t* yes);
/* Operand image of cat(src0, src1) (channel-blocked fp32 [N, c/8, hin, win, 8]) for dsg_conv_args.src_operand:
 * [piece 2][N][(c0+c1)/8][hin+2][win+2][8] fp16 -- piece 0 = fp16(v), piece 1 = fp16((v - piece 0) * 2^11) of
 * v = silu(x * scale + shift) (gn_scale_shift [N][c0+c1][2], silu as in the conv call) or, without gn_scale_shift,
 * v = x * 2^-e (src_bound*: the conv's range guard, NULL = none); a one-pixel zero border stands for the conv's padding. */
int dsg_conv_operand_bytes(int32_t n, int32_t c, int32_t hin, int32_t win, int32_t dtype, size_t* bytes);
int dsg_conv_operand_prepare(const float* src0, int32_t c0, const float* src1, int32_t c1, int32_t n, int32_t hin,
                             int32_t win, const float* gn_scale_shift, int32_t silu, const uint32_t* src_bound,
                             const uint32_t* src_bound1, void* operand, int32_t dtype, void* stream);

/* OIHW (checkpoint layout) -> engine layout [Cin][k*k][cout_total], written at
 * column offset cout_off (used to fuse to_q/to_k/to_v into one projection). nn.Linear weights
 * [out][in] are the k=1 case. */
int dsg_conv_weight_relayout(const float* w_oihw, float* dst, int32_t cout, int32_t cin, int32_t ksize,
                             int32_t cout_total, int32_t cout_off, void* stream);
/* OIHW (3x3 or 1x1 / Linear) -> [Cin/16][2][k*k][2][cout_total padded to 64][8] fp16: hi part and 2^11-scaled lo part of every weight,
 * so that w == hi + lo * 2^-11 to 2^-24 relative (fp32-equivalent contraction on the f16 MFMA, conv_h2.hip). */
int dsg_conv_weight_relayout_h2(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, int32_t ksize,
                                int32_t cout_total /* 0 = cout */, int32_t cout_off, void* stream);
/* Generic form of the four *_h2* packers below, for every dsg_dtype: OIHW fp32 -> the matrix-core operand image
 * [phase][K/16][pieces][taps][2][n_total padded to 64][8] of 16-bit values; pieces = 2 (hi, 2^11-scaled lo fp16) for DSG_F32,
 * 1 rounded value for DSG_BF16 / DSG_F16.  kind: 0 forward conv (dsg_conv_args.weight_h2); 1 folded up-sampler
 * (weight_h2_fold); 2 stride-2 conv over the space-to-depth image (weight_h2_s2); 3 data gradient (K = cout, N = cin,
 * taps reversed: weight_h2 of the conv dX = conv(dY, .)); 4 data gradient of a stride-2 conv as four 2x2 phase convs of
 * the low-resolution dY (weight_h2_fold of a call with upsample == 1: the adjoint of kind 2); 5 data gradient of an
 * up-sampler conv (nearest-2x + 3x3) as one stride-2 conv with a 4x4 window over the full-resolution dY (weight_h2_s2 of a
 * call with s2_window4 == 1: K = 4 cout in kind 2's (block, pixel parity) order, N = cin; cout % 8 == 0).  n_total / n_off place
 * this weight's N columns inside a wider matrix (kinds 0 and 3; 0 = no window). */
int dsg_conv_weight_pack(const float* w_oihw, void* dst, int32_t cout, int32_t cin, int32_t ksize, int32_t kind,
                         int32_t dtype, int32_t n_total, int32_t n_off, void* stream);
int dsg_conv_weight_pack_bytes(int32_t cout, int32_t cin, int32_t ksize, int32_t kind, int32_t dtype, int32_t n_total,
                               size_t* bytes);
/* The same for a whole table of weights in ONE launch: a training step re-packs every conv weight after the optimizer
 * step (the reference's loop: training_pipeline.py:84-91 -- optimizer.step() at :89 changes all of them), 140 calls of ~5 us
 * of work each in the mixed-precision tape.  A job is one dsg_conv_weight_pack call (same fields; n_pad = n_total, or cout
 * for kinds 0-2 / cin for kinds 3-4, rounded up to 64), or -- kind < 0 -- a plain copy of `cout` fp32 elements w -> dst.
 * The caller keeps two device arrays: the jobs and first[njobs + 1], the running sum of dsg_conv_weight_pack_batch_items
 * (host-only helper); total_items = first[njobs].  Same bits as the one-by-one calls. */
typedef struct {
  const float* w;
  void* dst;
  int32_t cout, cin, ksize, kind, dtype, n_total, n_off, n_pad;
} dsg_pack_job;
int dsg_conv_weight_pack_batch_items(const dsg_pack_job* job, int64_t* items);
int dsg_conv_weight_pack_batch(const dsg_pack_job* jobs_dev, const int64_t* first_dev, int32_t njobs, int64_t total_items,
                               void* stream);
/* OIHW 3x3 -> [phase 4][Cin/16][2][2x2 taps][2][cout padded to 64][8] fp16: Upsample2D (nearest x2) + this conv as
 * four 2x2 convs of the low-resolution input, one per output-pixel parity; taps that land on the same source pixel
 * are summed in fp32 before the split. */
int dsg_conv_weight_relayout_h2_fold(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, void* stream);
/* OIHW 3x3 -> [4 Cin/16][2][2x2 taps][2][cout padded to 64][8] fp16 for dsg_conv_args.weight_h2_s2 (cin % 8 == 0):
 * contraction index = (channel block, pixel parity (py, px), channel in block); zero where a (tap, parity) pair has
 * no 3x3 tap. */
int dsg_conv_weight_relayout_h2_s2(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, void* stream);
/* the same for the data-gradient conv (K = cout, N = cin padded to 64, taps reversed): [Cout/16][2][k*k][2][cin_pad][8] */
int dsg_conv_weight_relayout_h2_dgrad(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, int32_t ksize,
                                      void* stream);
/* OIHW -> [Cout][k*k flipped][cin_total]: the weight of the data-gradient convolution
 * dX = conv(dY, W^T flipped) (backward of training_pipeline.py:84 through :86). */
int dsg_conv_weight_relayout_dgrad(const float* w_oihw, float* dst, int32_t cout, int32_t cin, int32_t ksize,
                                   int32_t cin_total, int32_t cin_off, void* stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm (ResnetBlock2D.norm1/norm2, Attention.group_norm, conv_norm_out).
 * Statistics are per-channel (sum, sum of squares) in fp64 so that groups spanning the
 * [x || skip] concat of the up path combine exactly; finalize turns them into the per-(n, c)
 * scale/shift consumed by dsg_conv2d_fwd / dsg_gn_apply.
 * ---------------------------------------------------------------------------------------- */
int dsg_gn_channel_stats(const float* src0, int32_t c0, const float* src1, int32_t c1, int32_t n,
                         int32_t hw, double* chan_stats /* [N][c0+c1][2] */, void* stream);
/* the same statistics of one channel-blocked tensor [N][C/8][hw][8] (dsg_conv_args.dst_layout == 1), as `splits`
 * partial sums over equal runs of pixels (splits divides hw): feed dsg_gn_finalize_parts with tiles = splits */
int dsg_gn_channel_stats_blocked(const float* src, int32_t c, int32_t n, int32_t hw, int32_t splits,
                                 double* chan_stats /* [N][C][splits][2] */, void* stream);
/* bound[n] = max(bound[n], bits(sqrt(max over (c, tile) of the sum of squares))) from statistics [N][C][tiles][2]: the
 * dsg_conv_args.src_bound of a tensor whose statistics came from a pass of their own (bound zero-initialised) */
int dsg_range_bound_from_stats(const double* stats, int32_t n, int32_t c, int32_t tiles, uint32_t* bound, void* stream);
/* max |w| of a weight tensor (device scalar): the plan keeps a conv off the fp16x2 split when it leaves [2^-8, 3e4] */
int dsg_abs_max(const float* x, int64_t numel, float* out, void* stream);
/* the same for a channel-blocked tensor stored as `dtype` (dsg_dtype) */
int dsg_gn_channel_stats_blocked_dt(const void* src, int32_t c, int32_t n, int32_t hw, int32_t splits,
                                    double* chan_stats, int32_t dtype, void* stream);
int dsg_gn_finalize(const double* chan_stats, const float* gamma, const float* beta, int32_t n,
                    int32_t c, int32_t groups, int32_t hw, float eps,
                    float* scale_shift /* [N][C][2] */, void* stream);
/* finalize over cat(src0, src1) from per-tile partial statistics [N][c_i][tiles_i][2] (conv stats_out, or
 * dsg_gn_channel_stats output with tiles = 1); fixed summation order, fp64 */
int dsg_gn_finalize_parts(const double* stats0, int32_t c0, int32_t tiles0, const double* stats1, int32_t c1,
                          int32_t tiles1, const float* gamma, const float* beta, int32_t n, int32_t groups,
                          int32_t hw, float eps, float* scale_shift /* [N][c0+c1][2] */, void* stream);
/* the same, also leaving in bound[n] (zero-initialised by the caller; atomic max) the range bound of cat(src0, src1) for
 * dsg_conv_args.src_bound: the resnet's shortcut conv reads the tensors this norm normalises, un-normalised */
int dsg_gn_finalize_parts_bound(const double* stats0, int32_t c0, int32_t tiles0, const double* stats1, int32_t c1,
                                int32_t tiles1, const float* gamma, const float* beta, int32_t n, int32_t groups,
                                int32_t hw, float eps, float* scale_shift, uint32_t* bound, void* stream);
/* the same, also returning (mean, rstd) per (n, c) for the backward pass (training forward) */
int dsg_gn_finalize_parts_train(const double* stats0, int32_t c0, int32_t tiles0, const double* stats1, int32_t c1,
                                int32_t tiles1, const float* gamma, const float* beta, int32_t n, int32_t groups,
                                int32_t hw, float eps, float* scale_shift, float* mean_rstd, void* stream);
int dsg_gn_apply(const float* src, const float* scale_shift, int32_t silu, float* dst, int32_t n,
                 int32_t c, int32_t hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * Self-attention core (UNetMidBlock2D.attentions.0, Attn{Down,Up}Block2D): softmax(Q^T K / sqrt(d)) V
 * per (n, head) over L = H*W tokens.  qkv is the fused projection output [N][3*C][L] (q rows
 * first, then k, then v; channel = head*d + i); out is [N][C][L].
 * ---------------------------------------------------------------------------------------- */
int dsg_attention_fwd(const float* qkv, float* out, int32_t n, int32_t c, int32_t heads, int32_t l,
                      void* stream);
/* the same with the products in `dtype` (dsg_dtype): DSG_BF16 / DSG_F16 round q, k, v and the probabilities once and
 * issue ONE matrix-core product each (head_dim 8, L % 32 == 0; other shapes run the fp32 kernel); q/k/v/out stay fp32
 * [N, C, L], the scores, running maximum, denominators and accumulators fp32 (BASELINE configs[4]: "MFMA bf16 attn") */
int dsg_attention_fwd_dt(const float* qkv, float* out, int32_t n, int32_t c, int32_t heads, int32_t l,
                         int32_t dtype, void* stream);
/* The same with q, k, v and the output channel-blocked -- qkv [N][3C/8][L][8], out [N][C/8][L][8], elements fp32
 * (DSG_F32) or the 16-bit type of `dtype` -- for head_dim 8 (one channel block per head) and L % 32 == 0: the layout
 * dsg_unet_forward keeps its intermediates in, so the q/k/v projection in front and the out-projection behind move
 * whole channel blocks (anything else: DSG_ERR_UNSUPPORTED_SHAPE, use dsg_attention_fwd_dt on [N,3C,L] fp32). */
int dsg_attention_fwd_blocked(const void* qkv, void* out, int32_t n, int32_t c, int32_t heads, int32_t l, int32_t dtype,
                              void* stream);

/* ------------------------------------------------------------------------------------------
 * Timestep path (UNet2DModel.time_proj + time_embedding + every ResnetBlock2D.time_emb_proj):
 *   act = silu(linear_2(silu(linear_1(sinusoid(t)))))   [N][dim]   (dim = 4*block_out_channels[0])
 *   proj = act @ Wp^T + bp                               [N][proj_total]
 * timesteps: device int64 [N] (training_pipeline.py:76).  W1 [dim][ch], W2 [dim][dim], Wp [proj_total][dim].
 * freqs: device fp32 [ch/2] = exp(-ln(10000) * i / (ch/2)), host-computed (the angle t*freq is
 * ulp-sensitive at t ~ 1000, so the table is the caller's -- the shim builds it with the reference's ops).
 * ---------------------------------------------------------------------------------------- */
int dsg_time_embed_fwd(const int64_t* timesteps, const float* freqs, int32_t n, int32_t ch, int32_t dim,
                       const float* w1, const float* b1, const float* w2, const float* b2, float* act,
                       void* stream);
int dsg_linear_fwd(const float* x, const float* w, const float* b, float* y, int32_t n, int32_t in_f,
                   int32_t out_f, void* stream);

/* ------------------------------------------------------------------------------------------
 * Noise scheduler elementwise math (diffusers DDPMScheduler / DDIMScheduler):
 *   add_noise   training_pipeline.py:80       x_t = sqrt_a[n]*x0 + sqrt_1ma[n]*eps
 *   ddpm_step   DDPMPipeline loop (training_pipeline.py:26-32, generation.py:14-20)
 *   ddim_step   BASELINE.json configs[1], [3]
 * Host-computed fp32 scalars are passed by value; evaluation order matches the reference so that
 * results are bit-identical to the torch-CPU expression on the same inputs.
 * ---------------------------------------------------------------------------------------- */
int dsg_add_noise(const float* x0, const float* noise, const float* sqrt_a /* device [N] */,
                  const float* sqrt_1ma /* device [N] */, float* out, int32_t n, int64_t per_sample,
                  void* stream);
/* (`noise` is read once, coalesced: device memory, or device-accessible PINNED host memory -- the evaluate call's seeded CPU
 *  generator draws into a pinned buffer that the kernel reads in place over PCIe instead of a copy per step) */
int dsg_ddpm_step(const float* sample, const float* eps, const float* noise /* NULL when t == 0 */,
                  float* prev, int64_t numel, float sqrt_beta_prod_t, float sqrt_alpha_prod_t,
                  float clip /* <=0: no clip */, float coef_x0, float coef_xt, float sigma, void* stream);
int dsg_ddim_step(const float* sample, const float* eps, float* prev, int64_t numel,
                  float sqrt_beta_prod_t, float sqrt_alpha_prod_t, float clip, float sqrt_alpha_prev,
                  float dir_coef, void* stream);
/* Device address of a pinned host buffer, for the kernels that read host memory in place (dsg_ddpm_step's `noise`).  Host-only. */
int dsg_host_device_pointer(const void* host, void** device);
/* Counter-based device noise: Philox4x32-10 + Box-Muller (opt-in; the default training loop keeps the reference's host draw).
 *   replaces   training_pipeline.py:72   noise = torch.randn(batch.shape).to(device)      [a serial CPU draw + an H2D copy]
 *         and  training_pipeline.py:80   noisy = noise_scheduler.add_noise(batch, noise, t)  in the same pass
 * Element e of the flat tensor takes lane e % 4 of Philox4x32-10(counter = (e/4 lo, e/4 hi, offset lo, offset hi), key = (seed lo,
 * seed hi)); lanes (0,1) and (2,3) are Box-Muller pairs: u1 = (float(r0) + 0.5f) * 2^-32, u2 = float(r1) * 2^-32,
 * z0 = sqrt(-2 ln u1) cos(2 pi u2), z1 = ... sin(...).  The uint32 stream is bit-exact against oracle/philox_oracle.py and the
 * Random123 known-answer vectors; the normals agree with the fp64 evaluation of the same formulas to fp32 rounding.
 * Stateless: (seed, offset) names a tensor; the caller advances `offset` per draw.  `noisy` is bitwise
 * dsg_add_noise(x0, noise, ...) on the `noise` this call writes. */
int dsg_philox_u32(uint32_t* out, int64_t numel, uint64_t seed, uint64_t offset, void* stream);
int dsg_philox_normal(float* out, int64_t numel, uint64_t seed, uint64_t offset, void* stream);
int dsg_add_noise_philox(const float* x0, const float* sqrt_a /* device [N] */, const float* sqrt_1ma /* device [N] */,
                         float* noisy, float* noise, int32_t n, int64_t per_sample, uint64_t seed, uint64_t offset,
                         void* stream);
/* Pipeline post-process (DDPMPipeline.__call__ tail, App. A.4): (x/2+0.5).clamp(0,1), NCHW -> NHWC;
 * mode 0: float out; mode 1: uint8 round (generation.py `.images`); mode 2: uint8 truncation
 * (training_pipeline.py:21-22). */
int dsg_postprocess(const float* x, void* out, int32_t n, int32_t c, int32_t hw, int32_t mode, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole-network plan: UNet2DModel.forward as one call, for the sampler loops
 * (DDPMPipeline.__call__: training_pipeline.py:26-32, generation.py:14-20).
 * ---------------------------------------------------------------------------------------- */
typedef struct dsg_unet dsg_unet_t;

typedef struct {
  int32_t in_channels, out_channels;
  int32_t sample_h, sample_w;
  int32_t layers_per_block;
  int32_t num_blocks;               /* <= 8 */
  int32_t block_out_channels[8];
  int32_t down_attn[8];             /* 1: AttnDownBlock2D, 0: DownBlock2D */
  int32_t up_attn[8];               /* 1: AttnUpBlock2D,   0: UpBlock2D  */
  int32_t norm_num_groups;
  float norm_eps;
  int32_t attention_head_dim;
  int32_t add_attention;            /* mid-block attention */
  int32_t compute_dtype;            /* dsg_dtype: DSG_F32 = the fp32-equivalent engine; DSG_BF16 / DSG_F16 = the intermediate
                                       activations are 16-bit channel-blocked tensors and every conv / projection runs
                                       once on the bf16 / f16 matrix cores (x, eps, parameters, statistics stay fp32) */
  uint32_t flags;                   /* DSG_UNET_* bits below; 0 = defaults */
} dsg_unet_config;
/* dsg_unet_config.flags -- per-plan behaviour (nothing process-global):
 *   DSG_UNET_BATCH_INVARIANT  kernel selection never depends on the batch size: the small-batch split-K path (K contracted
 *                             in parallel slices when a layer's grid covers at most half the chip) is not taken, so row i
 *                             of a batch-B call is BITWISE the batch-1 call on row i.  Default off: batch-1 / batch-5
 *                             sampling (training_pipeline.py:26-32, generation.py:14-20) is ~20 % faster with the split,
 *                             and equal to the one-slice result to fp32 round-off (<= 2e-6 relative). */
#define DSG_UNET_BATCH_INVARIANT 1u

int dsg_unet_create(const dsg_unet_config* cfg, dsg_unet_t** out);
void dsg_unet_destroy(dsg_unet_t* h);
/* Copies (and re-lays-out) one checkpoint tensor, named by its diffusers state-dict key
 * (SURVEY App. A.5), from device memory into the plan.  Stream-asynchronous like every other entry
 * point: `data` must stay valid until the work queued on `stream` has run; no host synchronisation,
 * legal under stream capture.  In the fp32-equivalent mode a conv weight's max|w| (the range guard of
 * the fp16x2 split) is left in a device table of the plan; dsg_unet_commit_params reads the table back.
 * The extra key "time_proj.freqs" ([block_out_channels[0]/2]) overrides the sinusoid frequency
 * table (default: correctly rounded exp computed on the host at create time). */
int dsg_unet_set_param(dsg_unet_t* h, const char* name, const float* data, int64_t numel, void* stream);
/* Settles the range guard of every weight uploaded since the last commit: ONE device-to-host copy of
 * the maxima table and ONE synchronisation of the stream the uploads were queued on (replaces the
 * reference-side per-tensor `load_state_dict` walk of train.py:59 with a single hand-over point).
 * dsg_unet_workspace_bytes / dsg_unet_forward call it themselves when maxima are outstanding; call it
 * explicitly after a parameter refresh that is followed by stream capture (a synchronisation is
 * illegal inside a capture).  A no-op when nothing is outstanding. */
int dsg_unet_commit_params(dsg_unet_t* h);
/* Number of parameters the plan expects / that have been set. */
int dsg_unet_num_params(const dsg_unet_t* h, int64_t* expected_tensors, int64_t* set_tensors,
                        int64_t* total_elements);
int dsg_unet_param_name(const dsg_unet_t* h, int64_t index, const char** name, int64_t* numel);
int dsg_unet_workspace_bytes(dsg_unet_t* h, int32_t batch, size_t* bytes);
/* eps = UNet(x, t).  timesteps: device int64 [batch].  Stream-asynchronous and legal under stream capture once the plan's
 * parameters are settled (dsg_unet_commit_params; the first call of a shape also sizes the workspace on the host): a captured
 * denoising step -- this call + dsg_ddim_step -- replays bit-identically to the eager calls (tests/test_gpu_unet.py).  Nothing
 * in the library issues hipMemsetAsync on a caller's stream (a memset NODE was found un-ordered against the kernel nodes behind
 * it under hipGraph replay on ROCm 7.2: zero-fills are kernels). */
int dsg_unet_forward(dsg_unet_t* h, const float* x, const int64_t* timesteps, float* out, int32_t batch,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step (training_pipeline.py:70-97): backward of the layers above, loss, clip, optimizer.
 *   loss = F.mse_loss(model(noisy, t), noise)      :84-85     dsg_mse_loss
 *   accelerator.backward(loss)                      :86        dsg_conv2d_wgrad, dsg_conv2d_fwd with
 *                                                              dsg_conv_weight_relayout_dgrad weights,
 *                                                              dsg_gn_bwd, dsg_attention_bwd, dsg_linear_bwd,
 *                                                              dsg_silu_bwd, dsg_channel_sums
 *   accelerator.clip_grad_norm_(params, 1.0)        :88        dsg_l2_norm (+ dsg_clip_scale, or fused below)
 *   optimizer.step()  (torch.optim.AdamW, train.py:66)  :89    dsg_adamw_step
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const float* src0;            /* saved conv INPUT before norm/activation: [N, c0, hin, win] */
  const float* src1;            /* optional second source (channel concat) */
  int32_t c0, c1;
  int32_t n, hin, win;
  int32_t upsample;             /* 1: the conv ran on the nearest x2 upsampled source */
  int32_t ksize, stride;
  int32_t cout;
  const float* dy;              /* [N, dy_ctotal, hout, wout]; this conv's cout channels start at dy_coff */
  int32_t dy_ctotal, dy_coff;   /* dy_ctotal 0 = cout (fused q/k/v projections share one dqkv tensor) */
  const float* gn_scale_shift;  /* optional [N][c0+c1][2]: the activation is recomputed in the gather */
  int32_t silu;
  float* dw;                    /* [cout][c0+c1][k][k] (OIHW, checkpoint layout); ACCUMULATED into */
  int32_t force_direct;         /* test hook: VALU reference kernel */
  void* workspace;              /* split-K partial sums, summed in a fixed order (deterministic) */
  size_t workspace_bytes;       /* >= dsg_conv2d_wgrad_workspace_bytes(args) */
  int32_t compute_dtype;        /* dsg_dtype.  DSG_F32: fp32 [N, C, H, W] tensors as above; stride-1 3x3 calls with cin % 32 == 0,
                                   cout % 64 == 0, wout % 32 == 0, hout % 2 == 0 and 1x1 calls with cin % 64 == 0, cout % 64 == 0,
                                   wout % 32 == 0, hout % 2 == 0 run on the fp16x2 split (both operands split into two fp16
                                   pieces, three matrix-core products, fp32 accumulate: fp32-class accuracy), every other shape
                                   on the exact f32 matrix-core / VALU kernels.  DSG_BF16 / DSG_F16 (the mixed-
                                   precision training tape): src0 / src1 / dy are channel-blocked [N, C/8, H, W, 8] tensors of
                                   that type, products run once on the 16-bit matrix cores, dw stays fp32.  Served: 3x3
                                   stride-1 convs with (c0 + c1) % 64 == 0, c0 % 64 == 0 when c1 > 0, cout % 64 == 0,
                                   dy_ctotal % 8 == 0 and dy_coff % 64 == 0, wout % 32 == 0, hout % 2 == 0; the two SAMPLER
                                   convs with c1 == 0 on a full-resolution grid of a multiple of 64 columns and 4 rows --
                                   upsample = 1 (Upsample2D's conv: src0 is the LOW-resolution [N, C/8, hin, win, 8] tensor, read at
                                   (y >> 1, x >> 1); no nearest-x2 copy is made) and stride = 2 (Downsample2D's conv: dy is the
                                   [N, Co/8, hin/2, win/2, 8] gradient, taken as zero between its pixels; no zero-stuffed copy is
                                   made); anything else is DSG_ERR_UNSUPPORTED_SHAPE (convert with dsg_layout_convert_dt and use
                                   the fp32 form). */
  float* dy_sums;               /* optional; the 16-bit form, and the fp32 form where the split 3x3 kernel serves it (stride 1,
                                   cin % 32 == 0, cout % 64 == 0, wout % 32 == 0, hout % 2 == 0; DSG_ERR_INVALID_ARG
                                   otherwise): out[n * dy_sums_stride + co] = sum over pixels of dY[n][co] (this
                                   conv's cout channels), the bias / time-embedding gradient -- a by-product of the dY tiles
                                   the kernel stages anyway (no pass of its own over dY: dsg_channel_sums_blocked) */
  int32_t dy_sums_stride;       /* 0 = cout */
  float* dy_bias_grad;          /* optional, with dy_sums: dy_bias_grad[co] += sum over n of dy_sums[n][co] -- the conv's bias gradient
                                   (training_pipeline.py:86 accumulates it in .grad), from the pass that finishes dy_sums instead of a
                                   dsg_reduce_rows_add launch after every conv */
} dsg_conv_wgrad_args;
int dsg_conv2d_wgrad(const dsg_conv_wgrad_args* a, void* stream);
int dsg_conv2d_wgrad_workspace_bytes(const dsg_conv_wgrad_args* a, size_t* bytes);

/* GroupNorm(+SiLU) backward over cat(src0, src1).  dx = d/dx of silu?(gn(x)) . dy (+ add); dgamma/dbeta are
 * accumulated.  ws_s12: [N][C][2] doubles, ws_coef: [N][C][3] floats. */
int dsg_gn_finalize_train(const double* chan_stats, const float* gamma, const float* beta, int32_t n, int32_t c,
                          int32_t groups, int32_t hw, float eps, float* scale_shift, float* mean_rstd /* [N][C][2] */,
                          void* stream);
int dsg_gn_bwd(const float* src0, int32_t c0, const float* src1, int32_t c1, const float* dy,
               const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu, int32_t n,
               int32_t hw, int32_t groups, const float* add0, const float* add1, float* dx0, float* dx1,
               float* dgamma, float* dbeta, double* ws_s12, float* ws_coef, void* stream);
/* The streaming passes of the backward walk on the mixed-precision tape: channel-blocked [N, C/8, hw, 8] tensors of
 * `dtype` (DSG_BF16 / DSG_F16); fp32 arithmetic, fp64 sums, results rounded once.  ws_s12 of dsg_gn_bwd_blocked holds
 * [N][C][2] + [N][C][dsg_gn_bwd_blocked_splits(hw)][2] doubles. */
int dsg_gn_bwd_blocked(const void* src0, int32_t c0, const void* src1, int32_t c1, const void* dy,
                       const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu, int32_t n,
                       int32_t hw, int32_t groups, const void* add0, const void* add1, void* dx0, void* dx1,
                       float* dgamma, float* dbeta, double* ws_s12, float* ws_coef, int32_t dtype, void* stream);
/* The same with a second fan-in term for source 0 (dx0 += add0 + add0b in fp32, rounded once): a resnet input that is both a
 * skip connection and the residual of its own block has two incoming gradients when its GroupNorm's backward runs -- this
 * saves the pass that added them (6 per training step of the configs[4] network). */
int dsg_gn_bwd_blocked_add2(const void* src0, int32_t c0, const void* src1, int32_t c1, const void* dy,
                            const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu,
                            int32_t n, int32_t hw, int32_t groups, const void* add0, const void* add0b, const void* add1,
                            void* dx0, void* dx1, float* dgamma, float* dbeta, double* ws_s12, float* ws_coef,
                            int32_t dtype, void* stream);
/* The same two with the statistics pass replaced by the partials a data-gradient conv's epilogue wrote (dsg_conv_args.gnb_*):
 * parts [N][C][ntile][2] doubles = per-tile (sum du, sum du * x); everything else as above (ws_s12: [N][C][2] doubles suffice).
 * x is then read by the conv's epilogue and by the apply pass -- four tensor passes instead of five. */
int dsg_gn_bwd_parts(const float* src0, int32_t c0, const float* src1, int32_t c1, const float* dy,
                     const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu, int32_t n,
                     int32_t hw, int32_t groups, const float* add0, const float* add1, float* dx0, float* dx1,
                     float* dgamma, float* dbeta, double* ws_s12, float* ws_coef, const double* parts, int32_t ntile,
                     void* stream);
/* dsg_gn_bwd with a second fan-in term for source 0 (dx0 += add0 + add0b, summed in that order: the bits of the separate add pass
 * it replaces -- a resnet input that is both a skip connection and its block's residual) and, optionally (parts != NULL, ntile > 0),
 * the epilogue partials in place of the statistics pass. */
int dsg_gn_bwd_add2(const float* src0, int32_t c0, const float* src1, int32_t c1, const float* dy,
                    const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu, int32_t n,
                    int32_t hw, int32_t groups, const float* add0, const float* add0b, const float* add1, float* dx0,
                    float* dx1, float* dgamma, float* dbeta, double* ws_s12, float* ws_coef, const double* parts,
                    int32_t ntile, void* stream);
int dsg_gn_bwd_blocked_parts(const void* src0, int32_t c0, const void* src1, int32_t c1, const void* dy,
                             const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu,
                             int32_t n, int32_t hw, int32_t groups, const void* add0, const void* add0b, const void* add1,
                             void* dx0, void* dx1, float* dgamma, float* dbeta, double* ws_s12, float* ws_coef,
                             int32_t dtype, const double* parts, int32_t ntile, void* stream);
int dsg_gn_bwd_blocked_splits(int32_t hw);
int dsg_channel_sums_blocked(const void* x, int32_t n, int32_t c, int32_t hw, float* out_nc, int32_t out_stride,
                             int32_t dtype, void* stream);
int dsg_add_dt(const void* a, const void* b, int64_t numel, void* out, int32_t dtype, void* stream);
int dsg_upsample_nearest2x_blocked(const void* src, void* dst, int64_t planes /* N * C/8 */, int32_t h, int32_t w,
                                   int32_t dtype, void* stream);
int dsg_sumpool2x2_blocked(const void* src, const void* add, void* dst, int64_t planes, int32_t h /* of dst */,
                           int32_t w, int32_t dtype, void* stream);
/* out_nc[n*out_stride + c] = sum over hw of x[n][c][:] (bias / time-embedding gradients) */
int dsg_channel_sums(const float* x, int32_t n, int32_t c, int32_t hw, float* out_nc, int32_t out_stride, void* stream);
int dsg_add(const float* a, const float* b, int64_t numel, float* out, void* stream);
/* Upsample2D's nearest x2 materialised ([planes][h][w] -> [planes][2h][2w]) and its adjoint, the 2x2 sum-pool
 * ([planes][2h][2w] -> [planes][h][w], + add if non-NULL): the backward of the up-sampler conv (upsamplers.0.conv)
 * runs its weight / data gradient through the matrix-core kernels at full resolution between these two. */
int dsg_upsample_nearest2x(const float* src, float* dst, int64_t planes, int32_t h, int32_t w, void* stream);
int dsg_sumpool2x2(const float* src, const float* add, float* dst, int64_t planes, int32_t h, int32_t w, void* stream);
int dsg_time_embed_fwd_train(const int64_t* timesteps, const float* freqs, int32_t n, int32_t ch, int32_t dim,
                             const float* w1, const float* b1, const float* w2, const float* b2, float* act,
                             float* emb, float* z1, float* z2, void* stream);
int dsg_reduce_rows_add(const float* src, int32_t n, int32_t c, int32_t stride, float* dst, void* stream);
/* attention: forward that also returns the log2-domain log-sum-exp [N][heads][L], and the backward
 * (dqkv [N][3C][L]; dsum_ws [N][heads][L] scratch). */
int dsg_attention_fwd_train(const float* qkv, float* out, float* lse, int32_t n, int32_t c, int32_t heads, int32_t l,
                            void* stream);
int dsg_attention_fwd_train_dt(const float* qkv, float* out, float* lse, int32_t n, int32_t c, int32_t heads, int32_t l,
                               int32_t dtype /* DSG_BF16 / DSG_F16: the mixed-precision tapes' arithmetic, as dsg_attention_fwd_dt */, void* stream);
int dsg_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                      float* dsum_ws, int32_t n, int32_t c, int32_t heads, int32_t l, void* stream);
/* The same with the arithmetic of the mixed-precision tapes: dtype DSG_BF16 / DSG_F16 and head_dim 8, l % 32 == 0 run on the
 * matrix cores (q, k, v, dO, P and dS rounded once to the 16-bit type, fp32 scores and accumulators: torch.autocast's split for the
 * attention core of training_pipeline.py:84-86 under mixed_precision; DSG_F16: dO is multiplied by a power of two before it is
 * rounded and the results are divided by it, so dS stays in fp16's range with or without a loss scale -- the forward must have
 * been dsg_attention_fwd_train_dt with the same dtype); DSG_F32 and every other shape: the exact kernels of dsg_attention_bwd. */
int dsg_attention_bwd_dt(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                         float* dsum_ws, int32_t n, int32_t c, int32_t heads, int32_t l, int32_t dtype, void* stream);
/* y = x W^T + b backward: dw [out][in] and db [out] are accumulated (skipped when NULL), dx [n][in] written. */
int dsg_linear_bwd(const float* x, const float* w, const float* dy, int32_t dy_stride, int32_t n, int32_t in_f,
                   int32_t out_f, float* dw, float* db, float* dx, void* stream);
/* out = x * alpha_dev[0] * mult (alpha_dev may be NULL) */
int dsg_scale(const float* x, int64_t numel, const float* alpha_dev, float mult, float* out, void* stream);
int dsg_silu_fwd(const float* z, int64_t numel, float* y, void* stream);
int dsg_silu_bwd(const float* z, const float* dy, int64_t numel, float* dz, void* stream);
/* loss[0] = mean((pred-target)^2); dpred = grad_scale * 2 (pred-target) / numel (skipped when NULL).
 * ws: >= 2048 doubles. */
int dsg_mse_loss(const float* pred, const float* target, int64_t numel, float grad_scale, float* loss, float* dpred,
                 double* ws, size_t ws_bytes, void* stream);
int dsg_l2_norm(const float* x, int64_t numel, float* norm, double* ws, size_t ws_bytes, void* stream);
int dsg_clip_scale(float* g, int64_t numel, const float* total_norm, float max_norm, void* stream);
/* GradScaler.unscale_ (accelerate's fp16 path, training_pipeline.py:86-91 under train.py:24): g *= inv_scale in place;
 * *found_inf (device int32, caller-zeroed, OR-ed into) becomes 1 when any element is inf or nan. */
int dsg_unscale_check(float* g, int64_t numel, float inv_scale, int32_t* found_inf, void* stream);
/* One fused AdamW update of a flat parameter slab (torch.optim.AdamW semantics: decoupled decay, lerp, bias
 * corrections).  total_norm != NULL applies clip_grad_norm_'s factor min(1, max_norm/(norm+1e-6)) on the fly. */
int dsg_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, double lr,
                   double beta1, double beta2, double eps, double weight_decay, int64_t step,
                   const float* total_norm, float max_norm, void* stream);

/* ------------------------------------------------------------------------------------------
 * Host-side batch PNG decoder of the training feeder (row f1; NO device work, no stream): `threads` native threads decode
 * paths[0..n) into the rows of out[n][h][w][c] (uint8; the loader's PINNED staging buffer), status[i] = 0 or a per-file code
 * (1 cannot open, 2 not a PNG, 3 unsupported variant, 4 shape differs from (h, w, c), 5 damaged) -- the caller reads THOSE
 * files with PIL.   replaces   utils/datasets/dataset.py:43-45 `ToTensor(Image.open(f))` on the training thread
 *                              scripts/train.py:35 DataLoader(num_workers=0)
 * 8-bit non-interlaced grey / RGB / grey+alpha / RGBA; rows equal np.asarray(PIL.Image.open(f)) bit for bit.
 * ---------------------------------------------------------------------------------------- */
int dsg_png_probe(const char* path, int32_t* h, int32_t* w, int32_t* c);
int dsg_png_decode_batch(const char* const* paths, int32_t n, uint8_t* out, int32_t h, int32_t w, int32_t c,
                         int32_t threads, int32_t* status /* [n] */);

/* ------------------------------------------------------------------------------------------
 * Rows next to the path.
 *   dsg_resize_normalize_u8  Image_Dataset.__getitem__ (utils/datasets/dataset.py:21-24,43-45): decoded uint8
 *                            [N][Hs][Ws][C] -> ToTensor -> bilinear (align_corners=False, no antialias) ->
 *                            (x - mean) / std -> fp32 [N][C][Ho][Wo]
 *   dsg_resize_normalize_f32 the same for the dataset's .pkl branch (utils/datasets/dataset.py:37-41): `fig_tensor`
 *                            float [N][Hs][Ws][C] -> permute -> bilinear -> (x - mean) / std (no / 255)
 *   dsg_hist_u8              get_gray_image's per-channel histograms (vectorization/utils/image_utils.py:26-28):
 *                            hist[n][c][256] of uint8 [N][H*W][C]
 *   dsg_mask_lut_u8          its +-0.1 background mask (:40) and extract_agents' threshold
 *                            (vectorization/direct/extract_vehicles.py:141-147) as byte look-ups:
 *                            out[n][p] = lut[n][0][img[n][p][ch0]] && (ch1 < 0 || lut[n][1][img[n][p][ch1]]) ? on : off
 * ---------------------------------------------------------------------------------------- */
int dsg_resize_normalize_u8(const uint8_t* src, int32_t n, int32_t hs, int32_t ws, int32_t c, float* dst, int32_t ho,
                            int32_t wo, float mean, float std, void* stream);
int dsg_resize_normalize_f32(const float* src, int32_t n, int32_t hs, int32_t ws, int32_t c, float* dst, int32_t ho,
                             int32_t wo, float mean, float std, void* stream);
int dsg_hist_u8(const uint8_t* img, int32_t n, int32_t hw, int32_t c, uint32_t* hist, void* stream);
int dsg_mask_lut_u8(const uint8_t* img, int32_t n, int32_t hw, int32_t c, int32_t ch0, int32_t ch1, const uint8_t* lut,
                    uint8_t on_value, uint8_t off_value, uint8_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Direct scene rasteriser: the reference's matplotlib drawing of lane way-points / segments
 * (utils/datasets/rasterization.py:57-126) and agent rectangles (utils/datasets/visualization.py:283-296) as one
 * pass over antialiased oriented boxes in pixel space, composited in list order.
 *   boxes [nbox][9] = (cx, cy, ux, uy, hx, hy, r, g, b): centre, unit axis, half extents, colour
 *   out   [3][h][w] fp32, initialised to the background colour
 * ---------------------------------------------------------------------------------------- */
int dsg_rasterize_boxes(const float* boxes, int32_t nbox, float* out, int32_t h, int32_t w, float bg0, float bg1,
                        float bg2, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement plumbing (no reference counterpart): per-kernel-class HIP-event timing on the launch
 * stream, used by bench.py's roofline leg.  Classes: 0 conv3x3 stride-1, 1 conv3x3 on the nearest-x2
 * upsampled input, 2 conv3x3 stride-2, 3 conv1x1, 4 direct (VALU) conv, 5 conv weight-gradient, 6 / 7 / 8 conv3x3
 * stride-1 / conv3x3 upsampled / conv1x1 on the fp16x2-split matrix-core path, 9 its 3x3 weight gradient, 10 the 3x3 kernel's
 * two-workgroup instantiation, 11 conv_in.hip, 12 conv_out.hip, 13 / 14 the 3x3 kernel / its two-workgroup instantiation
 * with a fused resnet shortcut (FLOPs and bytes of both convs), 20 + c the 16-bit modes' class c.  FLOPs/bytes are the
 * algorithmic figures of each launch (2*MACs; input + weights + output once).
 * ---------------------------------------------------------------------------------------- */
int dsg_prof_enable(int32_t on); /* 1 start (clears), 0 stop (clears); 2 pause / 3 resume, keeping the records */
int dsg_prof_summary(int32_t kernel_class, double* total_ms, double* total_flops, double* total_bytes,
                     int64_t* launches);
int dsg_prof_dump(const char* csv_path);
/* Kernel-selection switches for A/B measurements and tests -- a TEST HOOK, not part of the product surface: the call is
 * refused (DSG_ERR_INVALID_ARG) unless the process environment has DSG_TESTING=1, so a production process cannot change
 * process-global state through this header; what a deployment may want to choose per plan is in dsg_unet_config.flags.
 * (Defaults in brackets; python: env DSG_TUNING="key=value,..." implies DSG_TESTING=1.)
 *   1  K-chunk of the fp32 conv kernel: [0 = by grid size] | 4 | 8
 *   2  fp16x2-split conv kernels: [1] | 0 = every contraction on the f32 MFMA
 *   3  rows per wave of the split conv kernel: [0 = by grid size: rounds of 256 workgroups] | 2 | 4 | 3 = by grid size, but never 16-row tiles
 *      below 256 of them (the rule before round 3's end)
 *   5  GroupNorm statistics from the producing conv's epilogue: [1] | 0 = a pass of their own
 *   6  waves per workgroup of the 16-row split conv: [4] | 8 (two per SIMD)
 *   7  3x3 weight gradient on the split path: [1] | 0 = f32 MFMA
 *   8  up-sampler convs folded into 2x2 phase convs: [1] | 0 = nearest-x2 gather
 *  10  conv_out (cout <= 4) on the VALU kernel: [1] | 0 = zero-padded matrix-core tile
 *  11  pointwise split convs as 8-row tiles, two workgroups per CU: [1] | 0 = the 3x3 kernel's geometry
 *  13  dsg_unet_forward keeps its intermediate activations channel-blocked [N,C/8,H,W,8]: [1] | 0 = [N,C,H,W]
 *      (query dsg_unet_workspace_bytes again after changing it)
 *  17  blocked 3x3 convs whose 8-row x 64-cout grid covers at most half the CUs (small batches) as 32-cout
 *      workgroups: [1] | 0
 *  16  blocked 3x3 convs with cin <= 128 as 32-cout x 8-row workgroups, two per CU: [0] | 1 | n > 1 = when the
 *      64-cout x 16-row grid has at least n workgroups (1 = 512); bit-identical results, measured slower
 *  18  16-bit modes: 3x3 convs with cout % 128 == 0 as 128-cout workgroups while the grid fills the chip: [1] | 0
 *  19  split-K for grids of at most half the CUs (needs dsg_conv_args.splitk_ws): [1] | 0
 *  25  dsg_unet_forward keeps q, k, v and the attention output channel-blocked (head_dim 8): [1] | 0
 *  23  resnet shortcuts fused into conv2's K loop (dsg_conv_args.sc_*): [1] | 0 (0: dsg_conv2d_fuses_shortcut answers no)
 *  30  16-bit pointwise weight gradients on a kernel of their own (tiles up to 128 ci x 128 co): [1] | 0 = the 3x3 kernel's
 *      one-tap instantiation (64 x 64 workgroups)
 *  34  split-K also for grids of 129 .. 170 workgroups with at least 24 K-chunks (batch-5 sampling at the 32 x 32 level): three
 *      slices = two rounds of a third of the work: [1] | 0 = only grids of at most half the CUs split
 *  36  round 5's additions to the rows-per-wave rule (key 3 = 0): 16-row tiles for the convs that split K in three slices (one
 *      round of 195 .. 255 workgroups instead of two), and the four-tap kernels (folded up-sampler) priced at 0.62 of a 16-row
 *      workgroup per 8-row one instead of 0.55: [1] | 0
 *  32  maps narrower than a 32-column tile (16 x 16, 8 x 8: the deepest levels of BASELINE configs[3]'s 512 x 512 network) also
 *      take split-K, the folded up-sampler kernel and the stride-2 space-to-depth kernel: [1] | 0 = one-slice plain kernel and
 *      the exact f32-MFMA kernels for them
 *  37  GroupNorm-backward statistics from the data-gradient conv's epilogue (dsg_conv_args.gnb_*): [1] | 0 = dsg_conv2d_gnb_supported
 *      answers no (the statistics pass runs) | 2 = only the 64-cout workgroups carry the epilogue | 3 = as 1, but a call whose two
 *      x tensors meet INSIDE a channel tile (at a multiple of 32 channels) loses the epilogue (round 6's first rule)
 *  38  fp32 attention backward (head_dim 8, L % 32 == 0) as fp16x2-split products on the matrix cores: [1] | 0 = the VALU kernels
 *  39  16-bit weight gradient of Upsample2D's conv in the folded form -- x's own map as the K grid, dY read as its space-to-depth
 *      image, the 2 x 2 taps a pixel parity reads: 16 products per low-resolution pixel instead of 36: [1] | 0 = nine taps at full
 *      resolution with x addressed at (y >> 1, x >> 1)
 *  41  16-bit data-gradient convs with the GroupNorm-backward epilogue (gnb_*) on 64-cout workgroups, two per CU, whatever the
 *      plain conv would take: [1] | 0 = 128-cout workgroups where cout % 128 == 0 and the grid fills the chip (round 6's first form)
 *  40  stride-2 3x3 convs of fp32 [N,C,H,W] tensors (weight_h2_s2 given: the fp32 training tape's down-samplers) on the
 *      space-to-depth kernel, as the channel-blocked ones: [1] | 0 = the exact f32 MFMA kernel
 *  31  fp32-equivalent 3x3 weight gradients with cout % 128 == 0 as 32 ci x 128 co workgroups (a wave keeps two co tiles, nine
 *      (tap, co tile) units on every wave; conv_wgrad_h2w_kernel): [1] | 0 = the 32 ci x 64 co workgroup everywhere
 *  29  16-bit 3x3 weight gradients with cout % 128 == 0 as 64 ci x 128 co workgroups (a wave keeps two co tiles, one
 *      workgroup per CU): [1] | 0 = 64 x 64 workgroups, two per CU
 *  20  fp32-equivalent 3x3 convs with cin <= 128 on channel-blocked tensors: 8-row tiles with ONE weight slab in LDS,
 *      two workgroups per CU (grids of at least 512 workgroups): [1] | 0
 *  21  conv_in (fp32 [N,C<=8,H,W] image -> channel-blocked result, 16 x 32 pixel tiles, cout % 32 == 0) on its own kernel
 *      with built-in operand scaling and GroupNorm statistics (csrc/conv_in.hip): [1] | 0 = the exact f32-MFMA kernel
 *  22  conv_out (normalised channel-blocked source of <= 64 channels -> fp32 [N,C<=8,H,W] image, 16 x 32 pixel tiles) on its
 *      own matrix-core kernel with per-output-channel weight scaling (csrc/conv_out.hip): [1] | 0 = the VALU / padded kernels
 *  15  stride-2 convs of channel-blocked tensors on the split path: [1] | 0 = the f32 MFMA kernel
 *  14  attention with head_dim 8 on the matrix cores (fp16x2 split): [1] | 0 = the VALU kernel */
int dsg_set_tuning(int32_t key, int32_t value);
/* A counter that advances with every accepted dsg_set_tuning call, whichever kernel family the key belongs to (0 in a
 * production process): host-side caches of kernel-selection answers (dsg_conv2d_fuses_shortcut, dsg_conv2d_takes_operand,
 * dsg_unet_workspace_bytes) key on it. */
int32_t dsg_tuning_epoch(void);

#ifdef __cplusplus
}
#endif
#endif /* DSG_H_ */
