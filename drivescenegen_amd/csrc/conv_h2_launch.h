// Launch logic of conv_h2_kernel, shared by the three translation units that instantiate it:
// conv_h2.hip (PREC 0: fp32 tensors, fp16x2-split products), conv_h2_bf16.hip (PREC 1), conv_h2_f16.hip (PREC 2).
#pragma once
#include "conv_h2_kernel.h"

// 16-bit modes, 16-row tiles: workgroups the kernel is compiled to fit per CU (2: half the register file each -- one's
// patch loads and output stores run under the other's MFMAs; A/B with tools/build_variant.sh)
#ifndef DSG_H16_NT4_OCC
#define DSG_H16_NT4_OCC 2   // measured on the configs[4] forward at batch 64: 31.9 -> 28.3 ms per step
#endif

namespace dsg {

// kernel-selection switches (dsg_set_tuning); defined in conv_h2.hip
struct H2Tuning {
  int enabled = 1;
  int waves = 4;        // 16-row tiles: 4 waves x 4 rows or 8 waves x 2 rows (tuning key 6)
  int fold = 1;         // folded up-sampler convs (key 8: A/B against the x2 gather)
  int stats = 1;        // epilogue GroupNorm statistics (key 5: A/B against the separate pass)
  int bm32 = 0;         // 32-cout x 8-row workgroups, two per CU, for the shallow levels (key 16; measured slower than
                        // the 64 x 16 geometry: 0.356 vs 0.302 ms at 64 channels / 256^2 -- off)
  int bm32_min = 512;   // ... when the 64-cout x 16-row grid has at least this many workgroups
  int bm32_small = 1;   // 32-cout workgroups for grids of at most half the CUs (key 17)
  int s2 = 1;           // stride-2 convs on the split path (key 15: A/B against the f32 MFMA kernel)
  int pw_occ2 = 1;      // pointwise convs: 8-row tiles compiled for two workgroups per CU (key 11)
  int rows = 0;         // rows per wave: 0 = by grid size, 2 | 4 forced (key 3)
  int bm128 = 1;        // 16-bit modes: 128-cout workgroups where the grid still fills the chip (key 18)
  int splitk = 1;       // split-K for grids of at most half the CUs, when the caller gives scratch (key 19)
  int ws2 = 1;          // fp32-equivalent 3x3 convs with cin <= 128: 8-row tiles, one weight slab, two workgroups per CU (key 20)
  int fuse_sc = 1;      // resnet shortcuts fused into conv2's K loop (key 23: A/B against the separate 1x1 kernel)
  int splitk_mid = 1;   // split-K also for grids of 129 .. 170 workgroups with K >= 24 chunks: 3 slices (key 34)
  int narrow = 1;       // maps narrower than a tile (16 x 16, 8 x 8) also take split-K, the folded up-sampler and the stride-2 kernel
                        // (key 32: 0 = one-slice plain kernel / exact f32 MFMA kernels for them, the rule before round 4)
  int pre = 1;          // pre-staged operand images for the layers with >= pre_min_ct cout tiles per patch (key 26)
  int pre_min_ct = 16;  // ... (key 27: the threshold.  Measured, profiles/r03_operand_ablation.txt: at 4 -- every conv of the 256- / 512-channel
                        // levels -- the convs gain 8.5 % and the prepare passes cost what they gain; at 16 only the folded up-samplers of
                        // those levels qualify, whose patch is staged by 16-32 workgroups)
  int gnb = 1;          // GroupNorm-backward statistics from the data-gradient conv's epilogue (key 37: A/B against the statistics pass)
  int s2_nchw = 1;      // stride-2 convs of fp32 [N,C,H,W] tensors on the space-to-depth kernel too (key 40: A/B against the exact f32 kernel)
  int gnb_bm64 = 1;     // 16-bit data-gradient convs with the GNB epilogue on 64-cout workgroups, two per CU (key 41: 0 = 128-cout ones where the plain conv takes them)
  int gnb_seam64 = 1;   // ... also where the two x tensors meet inside a channel tile, at a multiple of 32 channels (key 37 = 3: off)
  int rows_rule = 1;    // round 5's additions to the rows rule: 16-row tiles under three-slice split-K, 0.62 for the four-tap kernels (key 36)
  int epoch = 0;        // bumped by every change: plans key their cached workspace sizes on it
};
extern H2Tuning g_h2;
constexpr int H2_CUS = 256;

bool conv_h2_fold(const dsg_conv_args* a);
bool conv_h2_s2(const dsg_conv_args* a, int hout, int wout);
bool conv_h2_rows16(const dsg_conv_args* a, int hout, int wout);
bool splitk_prefers_bm64(int grid64, int nq);
// does the call take the fused-shortcut kernel (dsg_conv_args.sc_*)?  Shapes, layouts and dtype -- and, for a call that brings
// split-K scratch (splitk_ws), the slice count of that call, which is a function of the grid and therefore of the batch: a slice
// left with fewer shortcut chunks than the DMA ring is deep refuses the fusion.  "Row i of a batch == the batch-1 call on row i,
// bitwise" therefore holds for plans without split-K (DSG_UNET_BATCH_INVARIANT, or no splitk_ws); with split-K on, the
// fused / unfused choice -- a different summation order -- may differ between batch sizes (<= 2e-6 relative, tests/common.py)
bool conv_h2_sc_fusable(const dsg_conv_args* a, int hout, int wout);
// K slices (1 = no split) and statistics splits of the reduce pass for a call that may split (see dsg_conv_args.splitk_ws)
int conv_h2_splitk_slices(const dsg_conv_args* a, int hout, int wout, int* stat_splits);
// does the call's kernel read a pre-staged operand image (dsg_conv_args.src_operand)?  `wanted`: also apply the launcher's own
// pays-off rule (cout tiles per patch); without it the answer is "can", which is what a call that brings an image needs
bool conv_h2_takes_operand(const dsg_conv_args* a, int hout, int wout, bool wanted);
// GroupNorm-backward statistics from the epilogue (dsg_conv_args.gnb_*): does the kernel this call would launch have the GNB form,
// and does every channel tile of it lie in ONE of the two x tensors?  (tuning key 37 = 0: never -- the A/B switch)
bool conv_h2_gnb_ok(const dsg_conv_args* a, int hout, int wout);
// 16-bit modes: does the call take 128-cout workgroups, and with 16-row tiles?  (shared by the launcher and conv_h2_gnb_ok)
bool conv_h16_bm128(const dsg_conv_args* a, int hout, int wout, bool* r16);
int splitk_reduce_launch(const float* part, int slices, const dsg_conv_args* a, int hout, int wout, int stat_splits,
                         hipStream_t st);

template <int GM, int NT, int KS, int ACT, int NW = 4, int OCC = 1, int LAY = 0, int BM = 64, int PREC = 0, int WS = 0, int SC = 0, int PRE = 0, int GNB = 0>
static int h2_launch(dim3 grid, size_t lds, hipStream_t st, const ConvH2P& p) {
  auto kern = conv_h2_kernel<GM, NT, KS, ACT, NW, OCC, LAY, BM, PREC, WS, SC, PRE, GNB>;
  static bool raised = false;
  if (!raised) {
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    raised = true;
  }
  if constexpr (SC) {  // the shortcut phase's ring: 4 chunks of raw rows + 4 weight slabs (conv_h2_kernel.h)
    constexpr size_t SC_LDS = 4 * (size_t)((PREC ? 1 : 2) * NW * NT * 1024) + 4 * (size_t)(2 * (PREC ? 1 : 2) * BM / 64) * 1024;
    if (lds < SC_LDS) lds = SC_LDS;
  }
  // the epilogue's statistics tables live in the (by then free) K-loop buffers: [NW][row pairs][2][BM] partials + a
  // [values][64 lanes] table per wave; the pointwise kernels' buffers are smaller than that
  constexpr size_t STATS_LDS = ((size_t)NW * (NT / 2) * 2 * BM + (size_t)NW * (16 * (NT / 2) * 2) * 64) * sizeof(float);
  if (p.stats != nullptr && lds < STATS_LDS) lds = STATS_LDS;
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, p);
  return DSG_OK;
}

// PREC 0 serves every (layout, gather mode, geometry) combination conv_h2_eligible admits; the 16-bit modes serve
// channel-blocked tensors only (3x3: every tensor blocked, or blocked sources -> fp32 [N,C,H,W] result for conv_out;
// pointwise: any pair with at least one blocked side).
// The fp32-tape GNB instantiations live in a translation unit of their own (conv_h2_gnb.hip), built WITHOUT SLP vectorisation:
// in conv_h2.hip hipcc packed the epilogue's scalar fp32 statistics arithmetic into `v_pk_fma_f32 ... op_sel:[0,0,1]` -- the form
// that returns wrong values on lanes 48-63 next to a 16-deep MFMA of another wave (profiles/FINDINGS.md, round 4).  The two-rank
// and under-load tests caught it (overlapped vs deferred buckets no longer bitwise), tests/test_isa_policy.py names the kernels.
int conv_h2_gnb_f32_launch(bool nt4, dim3 grid, size_t lds, hipStream_t st, const ConvH2P& p);

template <int PREC>
int conv_h2_launch_t(const dsg_conv_args* a, int hout, int wout, hipStream_t st) {
  constexpr int NP = PREC ? 1 : 2;
  const int hout0 = hout, wout0 = wout;  // the conv's own output map (the kernel may re-tile it)
  // pointwise: 8-row tiles, two workgroups per CU (the only pointwise kernels that take channel-blocked tensors)
  const bool occ2 = a->ksize == 1 && (g_h2.pw_occ2 || a->src_layout || a->dst_layout);
  const bool nt4 = !occ2 && conv_h2_rows16(a, hout, wout);
  ConvH2P p;
  p.stats = a->stats_out;
  p.src0 = a->src0; p.src1 = a->src1; p.c0 = a->c0; p.c1 = a->c1; p.cin = a->c0 + a->c1;
  p.n = a->n; p.hin = a->hin; p.win = a->win;
  if (a->ksize == 1) {
    hout = hout * wout / H2_TW; wout = H2_TW;
    p.hin = hout; p.win = wout;
  }
  const bool fold = conv_h2_fold(a);
  const bool s2 = conv_h2_s2(a, hout, wout);
  if (fold) {  // the kernel tiles the LOW-resolution grid; outputs land at (2y + py, 2x + px)
    hout = a->hin;
    wout = a->win;
  }
  p.hc = (a->upsample && !fold) ? 2 * p.hin : p.hin;
  p.wc = (a->upsample && !fold) ? 2 * p.win : p.win;
  if (s2) {  // the patch lives on the output grid; its 4 C "channels" are (channel block, pixel parity) groups
    p.hc = hout;
    p.wc = wout;
    p.cin = 4 * a->c0;
  }
  p.hout = hout; p.wout = wout; p.cout = a->cout; p.cout_pad = (a->cout + 63) / 64 * 64;
  p.wh_stride = a->weight_h2_cout_stride ? a->weight_h2_cout_stride : p.cout_pad;
  p.wh = fold ? a->weight_h2_fold : (s2 ? a->weight_h2_s2 : a->weight_h2);
  p.bias = a->bias; p.ss = a->gn_scale_shift; p.silu = a->silu; p.temb = a->temb; p.temb_stride = a->temb_stride;
  p.res = a->residual; p.dst = a->dst;
  p.bound0 = (PREC == 0 && !a->gn_scale_shift) ? a->src_bound : nullptr;
  p.bound1 = (p.bound0 && a->src1) ? a->src_bound1 : nullptr;
  // fused shortcut: the 1x1 over the resnet's raw input rides on this conv2 (conv_h2_kernel's SC form)
  const bool sc = a->sc_weight_h2 != nullptr;
  p.sc_src0 = p.sc_src1 = p.sc_wh = nullptr; p.sc_bias = nullptr; p.sc_c0 = p.sc_c1 = p.sc_cin = p.sc_wh_stride = 0;
  p.pre = nullptr; p.pre_piece_stride = 0;
  const bool gnb = a->gnb_x0 != nullptr;
  p.gnb_x0 = a->gnb_x0; p.gnb_x1 = a->gnb_x1; p.gnb_c0 = a->gnb_x1 ? a->gnb_c0 : a->cout; p.gnb_ss = a->gnb_ss; p.gnb_silu = a->gnb_silu;
  if (gnb && !conv_h2_gnb_ok(a, hout, wout))
    return fail(DSG_ERR_UNSUPPORTED_SHAPE, "dsg_conv2d_fwd: gnb_* given for a call whose kernel has no GroupNorm-backward epilogue (ask dsg_conv2d_gnb_supported first)");
  if (sc) {
    if (!conv_h2_sc_fusable(a, hout, wout))
      return fail(DSG_ERR_UNSUPPORTED_SHAPE, "dsg_conv2d_fwd: sc_* given for a call that cannot fuse a shortcut (ask dsg_conv2d_fuses_shortcut first)");
    p.sc_src0 = a->sc_src0; p.sc_src1 = a->sc_src1; p.sc_c0 = a->sc_c0; p.sc_c1 = a->sc_src1 ? a->sc_c1 : 0;
    p.sc_cin = p.sc_c0 + p.sc_c1; p.sc_wh = a->sc_weight_h2; p.sc_wh_stride = (a->cout + 63) / 64 * 64; p.sc_bias = a->sc_bias;
    p.bound0 = PREC == 0 ? a->sc_src_bound : nullptr;
    p.bound1 = (p.bound0 && a->sc_src1) ? a->sc_src_bound1 : nullptr;
  }
  // 16-row tiles (NT = 4) when they still give every CU a workgroup; 8-row tiles otherwise
  const int th = nt4 ? 16 : 8;
  p.tiles_x = (wout + H2_TW - 1) / H2_TW; p.tiles_y = hout / th;
  const bool k1 = a->ksize == 1;
  const size_t lds = 2 * (size_t)(k1 ? (nt4 ? H2Geom<4, 1, 4, 1, 64, NP>::BUF_BYTES : H2Geom<2, 1, 4, 1, 64, NP>::BUF_BYTES)
                                     : (nt4 ? H2Geom<4, 3, 4, 9, 64, NP>::BUF_BYTES : H2Geom<2, 3, 4, 9, 64, NP>::BUF_BYTES)) +
                     (a->gn_scale_shift ? (size_t)p.cin * 2 * sizeof(float) : 0);  // + the scale/shift table
  dim3 grid(p.tiles_x * p.tiles_y * p.n * (p.cout_pad / H2_BM) * (fold ? 4 : 1));
  const int lay = (a->src_layout ? 1 : 0) | (a->dst_layout ? 2 : 0);
  const int act = a->gn_scale_shift ? (a->silu ? 2 : 3) : 0;
  int rc = DSG_OK;
  bool gnb_done = false;   // (a gnb call must end in a GNB instantiation: conv_h2_gnb_ok and this dispatch are checked against each other below)
  // split-K (PREC 0, every tensor channel-blocked, plain / stride-2 3x3 and pointwise): the slices write fp32 partials
  // to the caller's scratch, the reduce pass does what the epilogue would have
  int stat_splits = 1;
  const int slices = (PREC == 0 && a->splitk_ws) ? conv_h2_splitk_slices(a, hout0, wout0, &stat_splits) : 1;
  if (slices > 1) {
    const size_t slab = (size_t)p.n * p.cout * p.hout * p.wout * sizeof(float);
    if (a->splitk_ws_bytes < slab * slices)
      return fail(DSG_ERR_WORKSPACE_TOO_SMALL, "dsg_conv2d_fwd: splitk_ws %zu bytes < required %zu", a->splitk_ws_bytes, slab * slices);
    p.dst = a->splitk_ws;
    p.split_stride = slab;
    p.bias = nullptr; p.temb = nullptr; p.res = nullptr; p.stats = nullptr; p.sc_bias = nullptr;  // (the reduce pass adds them)
    grid.y = slices;
  } else {
    p.split_stride = 0;
  }
  // shallow levels of the fp32-equivalent path: 64 couts x 8 rows, ONE weight slab, two workgroups per CU (see WS)
  const bool ws2 = PREC == 0 && lay == 3 && g_h2.ws2 && !s2 && !fold && a->ksize == 3 && !a->upsample && p.cin <= 128 &&
                   wout % H2_TW == 0 && slices == 1 &&
                   (wout / H2_TW) * (hout / 8) * p.n * (p.cout_pad / H2_BM) >= 2 * H2_CUS;
  // pre-staged operand image: the 16-row kernels' PRE form (the image holds what the staging pass would have produced)
  const bool pre = a->src_operand != nullptr;
  if (pre) {
    if (PREC != 0 || !nt4 || ws2 || slices > 1 || !conv_h2_takes_operand(a, hout0, wout0, false))
      return fail(DSG_ERR_UNSUPPORTED_SHAPE, "dsg_conv2d_fwd: src_operand given for a call whose kernel stages its own patch (ask dsg_conv2d_takes_operand first)");
    p.pre = a->src_operand;
    p.pre_piece_stride = (size_t)p.n * p.cin * (p.hin + 2) * (p.win + 2) * 2;
  }
  int pi = -1;
  if (prof_on()) {
    const double px = (double)p.n * hout * wout * (fold ? 4 : 1);  // output pixels; FLOPs are the reference op's
    const int taps = a->ksize * a->ksize;
    const double cin_ref = a->c0 + a->c1;  // (stride 2: the kernel's 4 C x 4 taps are the reference op's C x 9)
    const double sc_cin = sc ? p.sc_cin : 0;  // fused shortcut: its 1x1 FLOPs, source and weight bytes count too
    const double es = (PREC && (lay & 1)) ? 2.0 : 4.0, ed = (PREC && (lay & 2)) ? 2.0 : 4.0, ew = PREC ? 2.0 : 4.0;
    // (the up-sampler's data gradient, s2_window4: the reference op is a 3x3 conv over the FULL-resolution map, 4 px output pixels' worth)
    pi = prof_begin((PREC ? 20 : 0) + (s2 ? 2 : (a->ksize == 1 ? 8 : (a->upsample ? 7 : (sc ? (ws2 ? 14 : 13) : (ws2 ? 10 : 6))))),
                    2.0 * px * (a->s2_window4 ? 4.0 : 1.0) * p.cout * (cin_ref * taps + sc_cin),
                    es * (double)p.n * (cin_ref + sc_cin) * p.hin * p.win + ew * (cin_ref * taps + sc_cin) * p.cout +
                        ed * px * p.cout * (p.res ? 2.0 : 1.0), st);
  }
#define DSG_H2_LAUNCH(GM, KS, ACT)                                                  \
  do {                                                                              \
    if (nt4 && g_h2.waves == 8) rc = h2_launch<GM, 2, KS, ACT, 8>(grid, lds, st, p); \
    else if (nt4) rc = h2_launch<GM, 4, KS, ACT>(grid, lds, st, p);                 \
    else rc = h2_launch<GM, 2, KS, ACT>(grid, lds, st, p);                          \
  } while (0)
#define DSG_H2_LAUNCH_BLK(GM, KS, ACT, LAY) /* channel-blocked: four-wave kernels only */ \
  do {                                                                              \
    if (nt4) rc = h2_launch<GM, 4, KS, ACT, 4, (PREC ? DSG_H16_NT4_OCC : 1), LAY, 64, PREC>(grid, lds, st, p);  \
    else rc = h2_launch<GM, 2, KS, ACT, 4, 1, LAY, 64, PREC>(grid, lds, st, p);      \
  } while (0)
#define DSG_H2_LAUNCH_PW(ACT) /* pointwise, two workgroups per CU: any layout pair */ \
  do {                                                                              \
    if (lay == 1) rc = h2_launch<0, 2, 1, ACT, 4, 2, 1, 64, PREC>(grid, lds, st, p); \
    else if (lay == 2) rc = h2_launch<0, 2, 1, ACT, 4, 2, 2, 64, PREC>(grid, lds, st, p); \
    else if (lay == 3) rc = h2_launch<0, 2, 1, ACT, 4, 2, 3, 64, PREC>(grid, lds, st, p); \
    else if constexpr (PREC == 0) rc = h2_launch<0, 2, 1, ACT, 4, 2, 0>(grid, lds, st, p); \
    else rc = fail(DSG_ERR_UNSUPPORTED_SHAPE, "dsg_conv2d_fwd: 16-bit pointwise conv needs a channel-blocked side"); \
  } while (0)
  // 32-cout workgroups: (a) optional, two per CU on the shallow levels (cin <= 128: LDS); (b) small batches: when
  // even the 8-row x 64-cout grid leaves more than half of the CUs idle, halve the cout tile to double the grid
  const bool bm32_ok = lay == 3 && !fold && !s2 && !k1 && !a->upsample && wout % H2_TW == 0 && (!sc || PREC == 0);
  const bool bm32 = bm32_ok && ((g_h2.bm32 && p.cin <= 128 && (int)grid.x >= g_h2.bm32_min) ||
                                (g_h2.bm32_small && !nt4 && (int)grid.x <= H2_CUS / 2 &&
                                 !(slices > 1 && splitk_prefers_bm64((int)grid.x, p.cin / H2_KC))));
  if (pre) {
    if constexpr (PREC == 0) {
      const size_t lpre = 2 * (size_t)(fold ? H2Geom<4, 3, 4, 4, 64, NP, 1>::BUF_BYTES : H2Geom<4, 3, 4, 9, 64, NP, 1>::BUF_BYTES);
      if (fold) rc = h2_launch<2, 4, 3, 0, 4, 1, 3, 64, 0, 0, 0, 1>(grid, lpre, st, p);
      else if (sc) rc = h2_launch<0, 4, 3, 0, 4, 1, 3, 64, 0, 0, 1, 1>(grid, lpre, st, p);
      else rc = h2_launch<0, 4, 3, 0, 4, 1, 3, 64, 0, 0, 0, 1>(grid, lpre, st, p);
    }
  } else if (s2 && a->s2_window4) {  // the up-sampler's data gradient: same patch and K axis, the taps of a 4x4 window
    if (lay) DSG_H2_LAUNCH_BLK(4, 3, 0, 3);
    else if constexpr (PREC == 0) {
      if (nt4) rc = h2_launch<4, 4, 3, 0>(grid, lds, st, p);
      else rc = h2_launch<4, 2, 3, 0>(grid, lds, st, p);
    }
  } else if (s2) {
    if (lay) DSG_H2_LAUNCH_BLK(3, 3, 0, 3);
    else if constexpr (PREC == 0) {
      if (nt4) rc = h2_launch<3, 4, 3, 0>(grid, lds, st, p);
      else rc = h2_launch<3, 2, 3, 0>(grid, lds, st, p);
    }
  } else if (fold) {
    if (lay) DSG_H2_LAUNCH_BLK(2, 3, 0, 3);
    else if constexpr (PREC == 0) {
      if (nt4) rc = h2_launch<2, 4, 3, 0>(grid, lds, st, p);
      else rc = h2_launch<2, 2, 3, 0>(grid, lds, st, p);
    }
  } else if (k1 && occ2) {
    if (act == 0) DSG_H2_LAUNCH_PW(0);
    else DSG_H2_LAUNCH_PW(3);
  } else if (k1) {
    if constexpr (PREC == 0) {
      if (act == 0) DSG_H2_LAUNCH(0, 1, 0);
      else DSG_H2_LAUNCH(0, 1, 3);
    }
  } else if (a->upsample) {
    if constexpr (PREC == 0) DSG_H2_LAUNCH(1, 3, 0);
  } else if (bm32) {
    // shallow levels: 32 couts x 8 rows x 32 columns per workgroup, two workgroups per CU
    const dim3 g32(((wout + H2_TW - 1) / H2_TW) * (hout / 8) * p.n * (p.cout_pad / 32), grid.y);
    const size_t lds32 = 2 * (size_t)H2Geom<2, 3, 4, 9, 32, NP>::BUF_BYTES + (a->gn_scale_shift ? (size_t)p.cin * 2 * sizeof(float) : 0);
    ConvH2P q = p;
    q.tiles_y = hout / 8;
    if (sc) {
      if constexpr (PREC == 0) rc = h2_launch<0, 2, 3, 2, 4, 2, 3, 32, 0, 0, 1>(g32, lds32, st, q);
    } else if (act == 0) rc = h2_launch<0, 2, 3, 0, 4, 2, 3, 32, PREC>(g32, lds32, st, q);
    else rc = h2_launch<0, 2, 3, 2, 4, 2, 3, 32, PREC>(g32, lds32, st, q);
  } else if (ws2 && !bm32) {
    // shallow levels: 64 couts x 8 rows, ONE weight slab (80 KB of LDS, half the register file): two workgroups per CU
    if constexpr (PREC == 0) {
      using GW = H2Geom<2, 3, 4, 9, 64, 2>;
      const dim3 gws(p.tiles_x * (hout / 8) * p.n * (p.cout_pad / H2_BM));
      const size_t ldsw = (size_t)GW::BUF_BYTES + GW::XHALFS * 2 + 64 + (a->gn_scale_shift ? (size_t)p.cin * 2 * sizeof(float) : 0);
      ConvH2P q = p;
      q.tiles_y = hout / 8;
      if (sc) rc = h2_launch<0, 2, 3, 2, 4, 2, 3, 64, 0, 1, 1>(gws, ldsw, st, q);
      else if (act == 0) rc = h2_launch<0, 2, 3, 0, 4, 2, 3, 64, 0, 1>(gws, ldsw, st, q);
      else rc = h2_launch<0, 2, 3, 2, 4, 2, 3, 64, 0, 1>(gws, ldsw, st, q);
    }
  } else if (lay == 3) {
    bool done128 = false;
    if constexpr (PREC != 0) {
      // 128-cout workgroups (four MFMA tiles per staged patch) while they still give every CU a workgroup
      bool r16 = false;
      if (conv_h16_bm128(a, hout, wout, &r16)) {
        const int per_row = p.tiles_x * p.n * (p.cout_pad / 128);
        const int th128 = r16 ? 16 : 8;
        const dim3 g128(per_row * (hout / th128));
        {
          ConvH2P q = p;
          q.tiles_y = hout / th128;
          const size_t ssb = a->gn_scale_shift ? (size_t)p.cin * 2 * sizeof(float) : 0;
          done128 = true;
          if (r16) {
            const size_t l128 = 2 * (size_t)H2Geom<4, 3, 4, 9, 128, NP>::BUF_BYTES + ssb;
            if (sc) rc = h2_launch<0, 4, 3, 2, 4, 1, 3, 128, PREC, 0, 1>(g128, l128, st, q);
            else if (gnb) { gnb_done = true; rc = h2_launch<0, 4, 3, 0, 4, 1, 3, 128, PREC, 0, 0, 0, 1>(g128, l128, st, q); }
            else if (act == 0) rc = h2_launch<0, 4, 3, 0, 4, 1, 3, 128, PREC>(g128, l128, st, q);
            else rc = h2_launch<0, 4, 3, 2, 4, 1, 3, 128, PREC>(g128, l128, st, q);
          } else {
            const size_t l128 = 2 * (size_t)H2Geom<2, 3, 4, 9, 128, NP>::BUF_BYTES + ssb;
            if (sc) rc = h2_launch<0, 2, 3, 2, 4, 1, 3, 128, PREC, 0, 1>(g128, l128, st, q);
            else if (gnb) { gnb_done = true; rc = h2_launch<0, 2, 3, 0, 4, 1, 3, 128, PREC, 0, 0, 0, 1>(g128, l128, st, q); }
            else if (act == 0) rc = h2_launch<0, 2, 3, 0, 4, 1, 3, 128, PREC>(g128, l128, st, q);
            else rc = h2_launch<0, 2, 3, 2, 4, 1, 3, 128, PREC>(g128, l128, st, q);
          }
        }
      }
    }
    if (!done128) {
      if (sc) {
        if (nt4) rc = h2_launch<0, 4, 3, 2, 4, (PREC ? DSG_H16_NT4_OCC : 1), 3, 64, PREC, 0, 1>(grid, lds, st, p);
        else rc = h2_launch<0, 2, 3, 2, 4, 1, 3, 64, PREC, 0, 1>(grid, lds, st, p);
      } else if (gnb) {
        if constexpr (PREC != 0) {
          gnb_done = true;
          if (nt4) rc = h2_launch<0, 4, 3, 0, 4, DSG_H16_NT4_OCC, 3, 64, PREC, 0, 0, 0, 1>(grid, lds, st, p);
          else rc = h2_launch<0, 2, 3, 0, 4, 1, 3, 64, PREC, 0, 0, 0, 1>(grid, lds, st, p);
        }
      } else if (act == 0) DSG_H2_LAUNCH_BLK(0, 3, 0, 3);
      else DSG_H2_LAUNCH_BLK(0, 3, 2, 3);
    }
  } else if (lay == 1) {  // 16-bit modes only (conv_out: blocked 16-bit sources -> fp32 [N,C,H,W])
    if constexpr (PREC != 0) DSG_H2_LAUNCH_BLK(0, 3, 2, 1);
  } else {
    if constexpr (PREC == 0) {
      if (gnb) { gnb_done = true; rc = conv_h2_gnb_f32_launch(nt4, grid, lds, st, p); }  // the fp32 tape's data-gradient conv ([N,C,H,W] both sides)
      else if (act == 0) DSG_H2_LAUNCH(0, 3, 0);
      else DSG_H2_LAUNCH(0, 3, 2);
    }
  }
#undef DSG_H2_LAUNCH
#undef DSG_H2_LAUNCH_BLK
#undef DSG_H2_LAUNCH_PW
  if (rc != DSG_OK) return rc;
  if (gnb && !gnb_done)   // (cannot happen while conv_h2_gnb_ok mirrors the dispatch above: the statistics table would hold the forward's sums)
    return fail(DSG_ERR_UNSUPPORTED_SHAPE, "dsg_conv2d_fwd: internal: a gnb_* call was dispatched to a kernel without the GroupNorm-backward epilogue");
  if (slices > 1) {
    DSG_LAUNCH_CHECK();
    rc = splitk_reduce_launch(static_cast<const float*>(a->splitk_ws), slices, a, hout0, wout0, stat_splits, st);
    if (rc != DSG_OK) return rc;
  }
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

}  // namespace dsg
