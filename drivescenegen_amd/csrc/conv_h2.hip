// fp16x2-split / bf16 / fp16 matrix-core convolution: host side (eligibility, tuning switches, weight packing).
// The kernel is conv_h2_kernel.h (see its header for the algorithm and the reference call sites); the launch logic is
// conv_h2_launch.h, instantiated here for the fp32-equivalent split (PREC 0) and in conv_h2_bf16.hip /
// conv_h2_f16.hip for the 16-bit modes (separate translation units: they compile in parallel).
#include "conv_h2_launch.h"

namespace dsg {

H2Tuning g_h2;

int conv_h2_launch_bf16(const dsg_conv_args* a, int hout, int wout, hipStream_t st);
int conv_h2_launch_f16(const dsg_conv_args* a, int hout, int wout, hipStream_t st);

// folded up-sampler mode: nearest x2 + 3x3 as four 2x2 convs of the low-resolution input (weight_h2_fold)
bool conv_h2_fold(const dsg_conv_args* a) {
  return g_h2.enabled && g_h2.fold && a->weight_h2_fold != nullptr && a->upsample == 1 && a->ksize == 3 && a->stride == 1 &&
         !a->pool2 && !a->gn_scale_shift && !a->weight_h2_cout_stride && (a->c0 + a->c1) % 16 == 0 &&
         (a->c1 == 0 || a->c0 % 16 == 0) && (a->win % H2_TW == 0 || (g_h2.narrow && (a->win == 16 || a->win == 8))) &&
         a->hin % 8 == 0 && a->cout % 8 == 0 &&
         (a->c0 + a->c1) <= 1024 && (a->compute_dtype == DSG_F32 || (a->src_layout == 1 && a->dst_layout == 1));
}

// stride-2 3x3 conv as a 2x2 conv over the space-to-depth image (GM = 3): channel-blocked tensors only
bool conv_h2_s2(const dsg_conv_args* a, int hout, int wout) {
  // (also on fp32 [N,C,H,W] tensors, the fp32 tape's layout: its down-sampler convs and its up-samplers' data gradient, s2_window4)
  const bool lay_ok = (a->src_layout == 1 && a->dst_layout == 1) ||
                      (a->src_layout == 0 && a->dst_layout == 0 && a->compute_dtype == DSG_F32 && (a->s2_window4 || g_h2.s2_nchw));
  return g_h2.enabled && g_h2.s2 && a->weight_h2_s2 != nullptr && a->stride == 2 && a->ksize == 3 && !a->upsample &&
         !a->pool2 && !a->gn_scale_shift && lay_ok && a->c1 == 0 && a->c0 % 8 == 0 &&
         a->cout % 8 == 0 && a->hin % 2 == 0 && a->win % 2 == 0 && hout % 8 == 0 &&
         (wout % H2_TW == 0 || (g_h2.narrow && (wout == 16 || wout == 8)));
}

bool conv_h2_eligible(const dsg_conv_args* a, int hout, int wout) {
  const int cin = a->c0 + a->c1;
  if (conv_h2_s2(a, hout, wout)) return true;
  if (a->stride != 1) return false;
  const int lay = (a->src_layout ? 1 : 0) | (a->dst_layout ? 2 : 0);
  const bool h16 = a->compute_dtype != DSG_F32;  // bf16 / fp16: the channel-blocked tensors are 16-bit
  // channel-blocked tensors: 3x3 stride-1 convs (plain or folded up-sampler) with every tensor blocked -- in the 16-bit
  // modes also blocked sources -> fp32 [N,C,H,W] (conv_out) --; pointwise convs with any pair of layouts
  if (h16 && lay == 0) return false;
  if (lay && a->ksize == 3) {
    const bool out_nchw = h16 && lay == 1 && !a->upsample && a->gn_scale_shift && a->silu;
    if (lay != 3 && !out_nchw) return false;
    if (a->upsample && !conv_h2_fold(a)) return false;
  }
  if (lay && (a->c0 % 8 || a->c1 % 8 || a->cout % 8)) return false;
  if (conv_h2_fold(a)) return true;
  if (a->weight_h2_cout_stride && (a->weight_h2_cout_stride % 64 || a->weight_h2_cout_stride < (a->cout + 63) / 64 * 64))
    return false;
  if (cin > 1024 && a->gn_scale_shift) return false;  // the GroupNorm scale/shift table shares LDS with the K-chunk buffers
  if (a->ksize == 1)  // pointwise: the map is re-tiled as (h*w/32) rows of 32 pixels, so only h*w matters
    return g_h2.enabled && a->weight_h2 != nullptr && a->stride == 1 && !a->upsample && !a->pool2 && !a->temb &&
           cin % 16 == 0 && (a->c1 == 0 || a->c0 % 16 == 0) && (hout * wout) % (8 * H2_TW) == 0 && a->cout % 8 == 0;
  if (a->gn_scale_shift && (!a->silu || a->upsample)) return false;  // combinations the U-Net does not have
  return g_h2.enabled && a->weight_h2 != nullptr && a->stride == 1 && a->upsample <= 1 && !a->pool2 &&
         cin % 16 == 0 && (a->c1 == 0 || a->c0 % 16 == 0) && (wout % H2_TW == 0 || wout == 16 || wout == 8) && (hout % 8 == 0) &&
         a->cout % 8 == 0;
}

// Fused shortcut (dsg_conv_args.sc_*): a resnet's conv2 -- 3x3, stride 1, GroupNorm + SiLU in front, every tensor
// channel-blocked, no residual -- takes the 1x1 conv_shortcut over the resnet's raw input into its own K loop.
// Every dsg_dtype; with split-K (small batches, fp32-equivalent mode) each K slice contracts its share of the shortcut too.
bool conv_h2_sc_fusable(const dsg_conv_args* a, int hout, int wout) {
  if (!g_h2.enabled || !g_h2.fuse_sc) return false;
  if (a->ksize != 3 || a->stride != 1 || a->upsample || a->pool2 || !a->gn_scale_shift || !a->silu || a->residual) return false;
  if (a->src_layout != 1 || a->dst_layout != 1 || a->weight_h2 == nullptr) return false;
  if (a->weight_h2_cout_stride && a->weight_h2_cout_stride != (a->cout + 63) / 64 * 64) return false;  // (no column windows)
  if (!conv_h2_eligible(a, hout, wout) || wout % H2_TW != 0) return false;
  const int cin = a->c0 + a->c1, sc_cin = a->sc_c0 + (a->sc_src1 ? a->sc_c1 : 0);
  if (cin < 2 * H2_KC || sc_cin < 4 * H2_KC || sc_cin % H2_KC || (a->sc_src1 && a->sc_c0 % H2_KC)) return false;  // (ring depth 4)
  if (a->splitk_ws) {  // split-K (small grids): every slice takes an equal share of the shortcut's chunks, at least the ring's depth
    const int slices = conv_h2_splitk_slices(a, hout, wout, nullptr);
    if (slices > 1) {
      const int ns = sc_cin / H2_KC, per = (ns + slices - 1) / slices;
      if (ns - (slices - 1) * per < 4) return false;
    }
  }
  return true;
}

// Pre-staged operand image (dsg_conv_args.src_operand, dsg_conv_operand_prepare): fp32-equivalent mode, every tensor
// channel-blocked, a stride-1 3x3 conv -- a resnet's conv1 / conv2 (GroupNorm + SiLU in front; with or without the fused
// shortcut) or the folded up-sampler conv (raw source) -- whose grid takes the 16-row kernel without split-K.  `wanted`
// adds the pays-off rule: the patch is staged by at least pre_min_ct workgroups (cout tiles x phases), i.e. the image
// replaces that many normalise + activate + split passes over it.
bool conv_h2_takes_operand(const dsg_conv_args* a, int hout, int wout, bool wanted) {
  if (!g_h2.enabled || !g_h2.pre || a->compute_dtype != DSG_F32) return false;
  if (a->ksize != 3 || a->stride != 1 || a->pool2 || a->src_layout != 1 || a->dst_layout != 1) return false;
  if (a->weight_h2_cout_stride) return false;
  const bool fold = conv_h2_fold(a);
  if (a->upsample && !fold) return false;
  if (!fold && (a->weight_h2 == nullptr || !a->gn_scale_shift || !a->silu)) return false;
  if (!conv_h2_eligible(a, hout, wout)) return false;
  if ((fold ? a->win : wout) % H2_TW != 0 || !conv_h2_rows16(a, hout, wout)) return false;
  if (a->splitk_ws && conv_h2_splitk_slices(a, hout, wout, nullptr) > 1) return false;
  const int cin = a->c0 + a->c1;
  if (g_h2.ws2 && !fold && cin <= 128) return false;  // (the two-workgroup kernel keeps its own staging)
  if (wanted && ((a->cout + 63) / 64) * (fold ? 4 : 1) < g_h2.pre_min_ct) return false;
  return true;
}

// tile geometry shared by the launcher and dsg_conv2d_stats_tiles
// 16-row tiles (4 rows per wave) are the efficient shape; 8-row tiles double the workgroup count.  The chip runs
// 256 workgroups at a time, so what counts is the number of ROUNDS: an 8-row workgroup costs ~0.55 of a 16-row one
// (half the MFMAs, the same fixed cost), and e.g. 320 workgroups of 16 rows (2 rounds) lose to 640 of 8 (3 x 0.55).
// cost8: what an 8-row workgroup costs relative to a 16-row one, in percent (half the MFMAs, the same fixed cost): 55 for the
// nine-tap kernels; the four-tap kernels (folded up-sampler, stride 2) have less MFMA work per tile to set against the same
// prologue / epilogue: 62 (batch-5 sampling, per-launch records under the forced geometries, profiles/r05_geometry_sweep.txt:
// the 256 -> 256 up-sampler conv at 5 x 640 tiles 148.3 us in 8-row tiles, 131.3 us in 16-row ones)
static bool rows16_pays(int b16, int cost8 = 55) {
  if (g_h2.rows == 2 || b16 <= 0) return false;
  if (g_h2.rows == 4) return true;
  // up to 128 tiles the 8-row grid still fits one round at 0.55 each.  129 .. 255 tiles go by the rounds rule as well: 160
  // workgroups of 16 rows on 62 % of the CUs beat 320 of 8 rows in two rounds (batch-5 sampling 6.03 -> 5.81 ms per step,
  // batch 3 4.05 -> 3.94, same-box A/B; the rule used to be "never below 256 tiles": key 3 = 3 keeps it for comparisons)
  if (g_h2.rows == 3 ? b16 < 256 : b16 <= 128) return false;
  const int r16 = (b16 + 255) / 256, r8 = (2 * b16 + 255) / 256;
  return 100 * r16 <= (g_h2.rows_rule ? cost8 : 55) * r8;
}

bool conv_h2_rows16(const dsg_conv_args* a, int hout, int wout) {
  if (conv_h2_fold(a)) {  // tiled on the low-resolution grid, four phases per cout tile
    const int cp = (a->cout + 63) / 64 * 64;
    return rows16_pays((a->hin % 16 == 0) ? (a->win / H2_TW) * (a->hin / 16) * a->n * (cp / H2_BM) * 4 : 0, 62);
  }
  if (a->ksize == 1) {
    hout = hout * wout / H2_TW;
    wout = H2_TW;
  }
  const int cout_pad = (a->cout + 63) / 64 * 64;
  const int b16 = (hout % 16 == 0) ? (wout / H2_TW) * (hout / 16) * a->n * (cout_pad / H2_BM) : 0;
  // Three-slice split-K (129 .. 170 eight-row tiles with long K: batch-5 sampling at the 32 x 32 level, generation.py:14-20):
  // 16-row tiles make that 3 x 65 .. 85 workgroups -- ONE round on 76-100 % of the chip, each workgroup with twice the MFMAs
  // per staged patch -- instead of two rounds of 8-row ones (per-launch records: 90.7 -> 84.2 us on the 512 -> 512 convs).
  if (g_h2.rows_rule && g_h2.rows == 0 && b16 > 0 && a->splitk_ws != nullptr && a->ksize == 3 && a->stride == 1) {
    int sp = 1;
    if (2 * b16 > H2_CUS / 2 && conv_h2_splitk_slices(a, hout, wout, &sp) == 3 && 3 * b16 <= H2_CUS) return true;
  }
  return rows16_pays(b16);
}

// Small grids with long K: 64-cout workgroups with MORE K slices instead of 32-cout workgroups with fewer -- when the 64-cout
// grid with the slices it may take (<= 4, >= 4 chunks each) fills at least three quarters of the chip.  Batch-1 sampling
// (training_pipeline.py:26-32) at the 64 x 64 level: 64 x 4 slices of 64 couts against 128 x 2 of 32 -- the same 256 workgroups,
// half the K chain, twice the MFMAs per staged patch (per-launch records, profiles/r05_geometry_sweep.txt: -5..-11 % on those convs).
bool splitk_prefers_bm64(int grid64, int nq) {
  if (!g_h2.rows_rule || !g_h2.splitk) return false;
  const int slices = std::min(4, std::min(H2_CUS / std::max(grid64, 1), nq / 4));
  // (long K only: a slice keeps at least six chunks -- with the 128 -> 128 convs' eight chunks cut in two the reduce pass costs
  //  more than the shorter chain saves: batch 1 measured +3.5 % without this condition)
  return slices >= 2 && nq >= 6 * slices && grid64 * slices >= 3 * H2_CUS / 4;
}

// Split-K plan of a call (see dsg_conv_args.splitk_ws): fp32 path, every tensor channel-blocked, plain or stride-2 3x3
// or pointwise; only when the tile grid the launcher would use covers at most half the CUs and K is long enough to cut.
int conv_h2_splitk_slices(const dsg_conv_args* a, int hout, int wout, int* stat_splits) {
  if (stat_splits) *stat_splits = 1;
  if (!g_h2.splitk || a->compute_dtype != DSG_F32 || a->src_layout != 1 || a->dst_layout != 1 || a->upsample) return 1;
  if (!conv_h2_eligible(a, hout, wout)) return 1;
  const bool s2 = conv_h2_s2(a, hout, wout);
  int th = hout, tw = wout;
  if (a->ksize == 1) {
    th = hout * wout / H2_TW;
    tw = H2_TW;
  }
  // maps narrower than a tile (16 x 16, 8 x 8: BASELINE configs[3]'s deepest levels -- 512 channels at 16 x 16 -- are ten
  // convs of a step on 64 workgroups at batch 8) split like the others: the slabs and the reduce pass are layout-only
  const bool narrow_ok = g_h2.narrow && a->ksize == 3 && (tw == 16 || tw == 8);
  if ((tw % H2_TW != 0 && !narrow_ok) || th % 8 != 0) return 1;
  const int cout_pad = (a->cout + 63) / 64 * 64;
  int grid = ((tw + H2_TW - 1) / H2_TW) * (th / 8) * a->n * (cout_pad / H2_BM);
  const int nq = (s2 ? 4 * a->c0 : a->c0 + a->c1) / H2_KC;
  if (a->ksize == 3 && !s2 && g_h2.bm32_small && grid <= H2_CUS / 2 && tw % H2_TW == 0 && !splitk_prefers_bm64(grid, nq))
    grid *= 2;  // (the launcher's 32-cout workgroups)
  if (nq < 8) return 1;
  int slices;
  if (grid > H2_CUS / 2) {
    // Grids of 129 .. 170 workgroups (batch-5 sampling -- generation.py:14-20 -- at the 32 x 32 level: 160) leave a third of the
    // chip idle for one long round (512 -> 512: 288 K-chunks per workgroup, 130 us at 0.22 of the roof).  Three slices make it two
    // rounds of a third of the work: 0.67 of the time by the rounds model, measured -2.2 % on the batch-5 step (6.25 -> 6.11 ms,
    // interleaved same-box passes).  Four slices for 171 .. 192 workgroups (three rounds of a quarter, 0.75 by the model) measured
    // +1.4 % at batch 3 and +2.7 % at batch 6: not taken.  Long K only (24 chunks); key 34.
    if (!g_h2.splitk_mid || nq < 24 || 3 * grid > 2 * H2_CUS) return 1;
    slices = 3;
  } else {
    slices = std::min(4, std::min(H2_CUS / grid, nq / 4));
  }
  while (slices > 1 && ((nq + slices - 1) / slices) * (slices - 1) >= nq) --slices;  // (no empty slice)
  if (slices < 2) return 1;
  if (stat_splits) {
    const int hw = hout * wout;
    int sp = 1;
    while (sp < 16 && hw % (2 * sp) == 0 && hw / (2 * sp) >= 2048) sp *= 2;
    *stat_splits = sp;
  }
  return slices;
}

// out = sum over the K slices (in slice order) + bias + temb + residual, channel-blocked fp32; grid = (C/8, n, stat
// splits): a thread owns a pixel's 8 channels; per-(n, c, split) (sum, sum of squares) for the GroupNorm that follows
__global__ __launch_bounds__(256) void splitk_reduce_blk_kernel(const float* __restrict__ part, int slices, size_t slab,
                                                                const float* __restrict__ bias, const float* __restrict__ bias2,
                                                                const float* __restrict__ temb, int temb_stride,
                                                                const float* __restrict__ res, float* __restrict__ dst,
                                                                int c, int hw_total, double* __restrict__ stats) {
  const int cb = blockIdx.x, n = blockIdx.y, sp = blockIdx.z, splits = gridDim.z;
  const int hw = hw_total / splits;
  const size_t base = ((size_t)n * c + cb * 8) * hw_total + (size_t)sp * hw * 8;
  float add[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    add[j] = (bias ? bias[cb * 8 + j] : 0.f) + (temb ? temb[(size_t)n * temb_stride + cb * 8 + j] : 0.f);
  if (bias2) {  // (the fused shortcut's bias: added after bias + temb, as the one-slice kernel's epilogue does)
#pragma unroll
    for (int j = 0; j < 8; ++j) add[j] += bias2[cb * 8 + j];
  }
  double s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.0;
  // Four pixels per thread and trip, every load of the trip issued before the first use (at most 4 slices: conv_h2_splitk_slices):
  // the maps this pass serves are small (64 .. 4096 pixels per image) and one pixel per trip made it a chain of hw / 256
  // dependent round trips -- 11 us per launch, 29 launches in a batch-1 step.  Same sums in the same order per thread.
  constexpr int RPX = 4, MAXS = 4;
  for (int i0 = threadIdx.x; i0 < hw; i0 += 256 * RPX) {
    float4 pa[RPX][MAXS][2], ra[RPX][2];
#pragma unroll
    for (int u = 0; u < RPX; ++u) {
      const int i = i0 + 256 * u;
      if (i < hw) {
        const size_t at = base + (size_t)i * 8;
#pragma unroll
        for (int k = 0; k < MAXS; ++k)
          if (k < slices) {
            pa[u][k][0] = *reinterpret_cast<const float4*>(part + k * slab + at);
            pa[u][k][1] = *reinterpret_cast<const float4*>(part + k * slab + at + 4);
          }
        if (res) {
          ra[u][0] = *reinterpret_cast<const float4*>(res + at);
          ra[u][1] = *reinterpret_cast<const float4*>(res + at + 4);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RPX; ++u) {
      const int i = i0 + 256 * u;
      if (i >= hw) break;
      const size_t at = base + (size_t)i * 8;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXS; ++k)
        if (k < slices) {
          const float4 a0 = pa[u][k][0], a1 = pa[u][k][1];
          v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w;
          v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
        }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += add[j];   // (the epilogue's order: accumulators + (bias + temb), then + residual)
      if (res) {
        const float4 r0 = ra[u][0], r1 = ra[u][1];
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
        v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      }
      *reinterpret_cast<float4*>(dst + at) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(dst + at + 4) = make_float4(v[4], v[5], v[6], v[7]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s[j] += (double)v[j];
        ss[j] += (double)v[j] * v[j];
      }
    }
  }
  if (stats == nullptr) return;
  __shared__ double red[16][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double x = s[j], y = ss[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      x += __shfl_down(x, o, 64);
      y += __shfl_down(y, o, 64);
    }
    if (lane == 0) {
      red[2 * j][wave] = x;
      red[2 * j + 1][wave] = y;
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int j = threadIdx.x >> 1, which = threadIdx.x & 1;
    stats[(((size_t)n * c + cb * 8 + j) * splits + sp) * 2 + which] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
  }
}

int splitk_reduce_launch(const float* part, int slices, const dsg_conv_args* a, int hout, int wout, int stat_splits,
                         hipStream_t st) {
  const int hw = hout * wout;
  const size_t slab = (size_t)a->n * a->cout * hw;
  hipLaunchKernelGGL(splitk_reduce_blk_kernel, dim3(a->cout / 8, a->n, a->stats_out ? stat_splits : 1), dim3(256), 0, st, part,
                     slices, slab, a->bias, a->sc_weight_h2 ? a->sc_bias : nullptr, a->temb, a->temb_stride, a->residual, a->dst,
                     a->cout, hw, a->stats_out);
  return DSG_OK;
}

bool conv_h16_bm128(const dsg_conv_args* a, int hout, int wout, bool* r16) {
  const int cout_pad = (a->cout + 63) / 64 * 64;
  if (!g_h2.bm128 || cout_pad % 128 != 0 || wout % H2_TW != 0) return false;
  // GroupNorm-backward epilogue over cat(x0, x1) whose seam is not a multiple of 128 channels (the 64 + 64 concat of the outermost
  // up block): 64-cout workgroups, two per CU.  The 128-cout tile CAN read x across the seam (the epilogue picks the tensor per
  // 32-channel slab), but on these short-K convs its epilogue is not hidden behind anything: bf16 B=128 step 180.4 ms against
  // 179.8 with the statistics pass, and 178.2 against 178.8 on 64-cout workgroups (profiles/r06_gnb_seam_ab.txt)
  if (a->gnb_x0 != nullptr && a->gnb_x1 != nullptr && a->gnb_c0 % 128 != 0 && g_h2.gnb_seam64) return false;
  // ... and so do ALL calls with the epilogue (key 41): one 128-cout workgroup per CU runs its VALU-bound epilogue with nothing beside
  // it, two 64-cout workgroups run it under each other's K loops -- bf16 B=128 step 175.7-175.9 ms against 177.3-177.5 with the
  // 128-cout workgroups (only the convs with at most 128 / 256 dY channels: 176.4-176.6 / 175.8-176.5; profiles/r06_gnb_bm64_ab.txt)
  if (a->gnb_x0 != nullptr && g_h2.gnb_bm64) return false;
  const int per_row = (wout / H2_TW) * a->n * (cout_pad / 128);
  *r16 = hout % 16 == 0 && per_row * (hout / 16) >= H2_CUS;
  return per_row * (hout / (*r16 ? 16 : 8)) >= H2_CUS;
}

bool conv_h2_gnb_ok(const dsg_conv_args* a, int hout, int wout) {
  if (!g_h2.enabled || !g_h2.gnb || !g_h2.stats || !conv_h2_eligible(a, hout, wout)) return false;
  if (a->ksize != 3 || a->stride != 1 || a->upsample || a->pool2 || a->gn_scale_shift || a->sc_weight_h2 || a->src_operand ||
      a->residual || a->weight_h2 == nullptr || wout % H2_TW != 0 || hout % 8 != 0 || a->gnb_ss == nullptr)
    return false;
  const int lay = (a->src_layout ? 1 : 0) | (a->dst_layout ? 2 : 0);
  const bool h16 = a->compute_dtype != DSG_F32;
  if (h16 ? lay != 3 : lay != 0) return false;
  if (g_h2.waves == 8) return false;  // (the eight-wave A/B geometry has no GNB instantiation)
  int bm = H2_BM;
  if (h16) {
    bool r16 = false;
    if (conv_h16_bm128(a, hout, wout, &r16)) {
      bm = 128;
      if (g_h2.gnb == 2) return false;  // (A/B mode: only the two-workgroups-per-CU 64-cout kernels, whose epilogue runs under the other workgroup's K loop)
    } else {  // the launcher's 32-cout workgroups for small grids have no GNB form
      const int cout_pad = (a->cout + 63) / 64 * 64;
      const bool nt4 = conv_h2_rows16(a, hout, wout);
      const int grid = (wout / H2_TW) * (hout / (nt4 ? 16 : 8)) * a->n * (cout_pad / H2_BM);
      if (g_h2.bm32_small && !nt4 && grid <= H2_CUS / 2) return false;
      if (g_h2.bm32 && a->c0 + a->c1 <= 128 && grid >= g_h2.bm32_min) return false;
    }
  }
  // the two x tensors meet between two 32-channel slabs of a tile (the epilogue reads x slab by slab: the 64 + 64 concat of the
  // outermost up block under 128-cout workgroups; key 37 = 3: only between two tiles, round 6's first rule -- such calls then lose
  // the epilogue to a statistics pass over x and dA per half, 2 x 0.5 ms per conv at batch 128)
  if (a->gnb_x1 != nullptr && (a->gnb_c0 <= 0 || a->gnb_c0 >= a->cout || a->gnb_c0 % (g_h2.gnb_seam64 ? 32 : bm) != 0)) return false;
  return true;
}

int conv_h2_stats_tiles(const dsg_conv_args* a, int hout, int wout) {
  if (!g_h2.stats || !conv_h2_eligible(a, hout, wout)) return 0;
  if (a->splitk_ws) {  // the split-K path's reduce pass writes its own (coarser) partials
    int sp = 1;
    if (conv_h2_splitk_slices(a, hout, wout, &sp) > 1) return sp;
  }
  if (a->ksize == 1) return hout * wout / (8 * H2_TW);  // (pointwise: the map is re-tiled as rows of 32 pixels)
  // the folded up-sampler tiles the LOW-resolution grid, four phases per tile (the same count as below while win % 32 == 0;
  // a 16- or 8-column source still has one -- partly masked -- tile per 8 rows and phase)
  if (conv_h2_fold(a)) return 4 * (a->hin / 8) * ((a->win + H2_TW - 1) / H2_TW);
  return (hout / 8) * ((wout + H2_TW - 1) / H2_TW);  // 8-row x 32-column statistics tiles for either block height
}

int conv_h2_launch(const dsg_conv_args* a, int hout, int wout, hipStream_t st) {
  if (a->compute_dtype == DSG_BF16) return conv_h2_launch_bf16(a, hout, wout, st);
  if (a->compute_dtype == DSG_F16) return conv_h2_launch_f16(a, hout, wout, st);
  return conv_h2_launch_t<0>(a, hout, wout, st);
}

void conv_h2_set_enabled(int on) { g_h2.enabled = on; ++g_h2.epoch; }
void conv_h2_set_rows(int r) { g_h2.rows = r; ++g_h2.epoch; }
void conv_h2_set_stats(int on) { g_h2.stats = on; ++g_h2.epoch; }
void conv_h2_set_fold(int on) { g_h2.fold = on; ++g_h2.epoch; }
void conv_h2_set_waves(int w) { g_h2.waves = w; ++g_h2.epoch; }
void conv_h2_set_pw_occ2(int v) { g_h2.pw_occ2 = v; ++g_h2.epoch; }
void conv_h2_set_s2(int v) { g_h2.s2 = v; ++g_h2.epoch; }
void conv_h2_set_bm32_small(int v) { g_h2.bm32_small = v; ++g_h2.epoch; }
void conv_h2_set_bm32(int v) { g_h2.bm32 = v != 0; if (v > 1) g_h2.bm32_min = v; ++g_h2.epoch; }
void conv_h2_set_bm128(int v) { g_h2.bm128 = v; ++g_h2.epoch; }
void conv_h2_set_splitk(int v) { g_h2.splitk = v; ++g_h2.epoch; }
void conv_h2_set_ws2(int v) { g_h2.ws2 = v; ++g_h2.epoch; }
void conv_h2_set_fuse_sc(int v) { g_h2.fuse_sc = v; ++g_h2.epoch; }
void conv_h2_set_pre(int v) { g_h2.pre = v; ++g_h2.epoch; }
void conv_h2_set_narrow(int v) { g_h2.narrow = v; ++g_h2.epoch; }
void conv_h2_set_splitk_mid(int v) { g_h2.splitk_mid = v; ++g_h2.epoch; }
void conv_h2_set_rows_rule(int v) { g_h2.rows_rule = v; ++g_h2.epoch; }
void conv_h2_set_gnb_bm64(int v) { g_h2.gnb_bm64 = v; ++g_h2.epoch; }
void conv_h2_set_s2_nchw(int v) { g_h2.s2_nchw = v; ++g_h2.epoch; }
void conv_h2_set_gnb(int v) { g_h2.gnb = v == 3 ? 1 : v; g_h2.gnb_seam64 = v != 3; ++g_h2.epoch; }
void conv_h2_set_pre_min_ct(int v) { g_h2.pre_min_ct = v > 0 ? v : 1; ++g_h2.epoch; }
int conv_h2_get_fuse_sc() { return g_h2.fuse_sc; }
int conv_in_tuning_epoch();  // conv_in.hip: its on/off switch moves the plan's statistics buffers too
int conv_h2_tuning_epoch() { return g_h2.epoch + conv_in_tuning_epoch(); }

// ---------------------------------------------------------------------------------------------------------------
// Weight packing: OIHW fp32 (the checkpoint layout) -> the kernel's operand image
//   [phase][K/16][piece][tap][k-group 2][n_pad][8]  16-bit values
// piece = (hi, 2^11-scaled lo) fp16 for DSG_F32 (w == hi + lo * 2^-11 to 2^-24 relative), one rounded value for
// DSG_BF16 / DSG_F16.  One thread per (phase, chunk, tap, g, n, j); `kind` selects what the (K, tap, N) axes mean:
//   0 forward conv        K = cin,  N = cout, taps k*k
//   1 folded up-sampler   K = cin,  N = cout, 4 phases x 2x2 taps: nearest x2 + 3x3 as four 2x2 convs of the
//                         low-resolution map; for output phase (py, px) the taps are sums of the 3x3 taps that land on
//                         the same source pixel -- rows {0 | 1+2} for py = 0, {0+1 | 2} for py = 1, the same in x
//                         (summed in fp32, dy outer / dx inner, then rounded / split)
//   2 stride-2 conv       K = 4 cin (channel block, pixel parity (py, px), channel in block), N = cout, 2x2 taps over
//                         the space-to-depth image: tap (ty, tx) of parity (py, px) is the 3x3 tap
//                         (2 ty + py - 1, 2 tx + px - 1) where that exists, zero where it does not
//   3 data gradient       K = cout, N = cin, taps reversed: dX = conv(dY, W^T flipped)
//   4 data gradient of a stride-2 conv, in the folded up-sampler's form: K = cout, N = cin, 4 phases x 2x2 taps over
//     the LOW-resolution dY; input pixel 2m + py receives dY[m] * W[1] (py = 0) or dY[m] * W[2] + dY[m+1] * W[0]
//     (py = 1), i.e. corner row tr of phase py is the 3x3 row {-, 1} / {2, 0}; the same in x
//   5 data gradient of the up-sampler (nearest-2x + 3x3 conv), as ONE stride-2 conv over the space-to-depth image of the
//     full-resolution dY (kind 2's axes with the roles of the channels swapped): K = 4 cout (channel block, pixel parity,
//     channel in block), N = cin, 2x2 taps; dX[y] = sum over the 4x4 window of dY rows 2y - 1 + i, i = 0..3, each row
//     weighted by the sum of the 3x3 rows that read low-resolution row y through it: {2}, {1, 2}, {0, 1}, {0}; tap ty
//     of parity py is window row i = 2 ty - py + 1; the same in x
// ---------------------------------------------------------------------------------------------------------------
// element i of the operand image of one weight (the body of weight_pack_kernel; weight_pack_batch_kernel runs it too)
__device__ __forceinline__ void weight_pack_elem(const float* __restrict__ w, unsigned short* __restrict__ dst, int cout_w,
                                                 int cin_w, int ksize, int kind, int dt, int n_pad, int n_off, int64_t i) {
  const int taps_w = ksize * ksize;
  const int taps = (kind == 1 || kind == 2 || kind == 4 || kind == 5) ? 4 : taps_w;
  const int kdim = kind == 2 ? 4 * cin_w : (kind == 5 ? 4 * cout_w : ((kind == 3 || kind == 4) ? cout_w : cin_w));
  const int ndim = (kind == 3 || kind == 4 || kind == 5) ? cin_w : cout_w;
  const int nq = kdim / 16, np = dt == 0 ? 2 : 1;
  {
    const int j = (int)(i % 8);
    int64_t r = i / 8;
    const int nn = (int)(r % ndim);
    r /= ndim;
    const int g = (int)(r % 2);
    r /= 2;
    const int tap = (int)(r % taps);
    r /= taps;
    const int q = (int)(r % nq);
    const int phase = (int)(r / nq);
    const int kk = q * 16 + g * 8 + j;  // index along the contraction axis
    float v = 0.f;
    if (kind == 0) {
      v = w[((int64_t)nn * cin_w + kk) * taps_w + tap];
    } else if (kind == 3) {
      v = w[((int64_t)kk * cin_w + nn) * taps_w + (taps_w - 1 - tap)];
    } else if (kind == 1) {
      const int py = phase >> 1, px = phase & 1, tr = tap >> 1, tc = tap & 1;
      const int dy0 = py == 0 ? (tr == 0 ? 0 : 1) : (tr == 0 ? 0 : 2), dy1 = py == 0 ? (tr == 0 ? 0 : 2) : (tr == 0 ? 1 : 2);
      const int dx0 = px == 0 ? (tc == 0 ? 0 : 1) : (tc == 0 ? 0 : 2), dx1 = px == 0 ? (tc == 0 ? 0 : 2) : (tc == 0 ? 1 : 2);
      const float* wp = w + ((int64_t)nn * cin_w + kk) * 9;
      for (int dy = dy0; dy <= dy1; ++dy)
        for (int dx = dx0; dx <= dx1; ++dx) v += wp[dy * 3 + dx];
    } else if (kind == 2) {
      const int gi = 2 * q + g, cb = gi >> 2, pp = gi & 3;
      const int dy = 2 * (tap >> 1) + (pp >> 1) - 1, dx = 2 * (tap & 1) + (pp & 1) - 1;
      if (dy >= 0 && dx >= 0) v = w[((int64_t)nn * cin_w + cb * 8 + j) * 9 + dy * 3 + dx];  // (dy, dx <= 2)
    } else if (kind == 5) {
      const int gi = 2 * q + g, cb = gi >> 2, pp = gi & 3;
      const int wi = 2 * (tap >> 1) - (pp >> 1) + 1, wj = 2 * (tap & 1) - (pp & 1) + 1;  // window row / column, 0..3
      const int dy0 = wi == 0 ? 2 : (wi == 1 ? 1 : 0), dy1 = wi <= 1 ? 2 : (wi == 2 ? 1 : 0);
      const int dx0 = wj == 0 ? 2 : (wj == 1 ? 1 : 0), dx1 = wj <= 1 ? 2 : (wj == 2 ? 1 : 0);
      const float* wp = w + ((int64_t)(cb * 8 + j) * cin_w + nn) * 9;
      for (int dy = dy0; dy <= dy1; ++dy)
        for (int dx = dx0; dx <= dx1; ++dx) v += wp[dy * 3 + dx];
    } else {  // kind 4
      const int py = phase >> 1, px = phase & 1, tr = tap >> 1, tc = tap & 1;
      const int ky = py == 0 ? (tr == 1 ? 1 : -1) : (tr == 0 ? 2 : 0);
      const int kx = px == 0 ? (tc == 1 ? 1 : -1) : (tc == 0 ? 2 : 0);
      if (ky >= 0 && kx >= 0) v = w[((int64_t)kk * cin_w + nn) * 9 + ky * 3 + kx];
    }
    const int64_t pb = (int64_t)phase * nq + q;
    const int64_t at = ((((pb * np + 0) * taps + tap) * 2 + g) * n_pad + n_off + nn) * 8 + j;
    if (dt == 0) {
      const _Float16 h1 = (_Float16)v;
      const _Float16 h2 = (_Float16)((v - (float)h1) * 2048.0f);
      dst[at] = __builtin_bit_cast(unsigned short, h1);
      dst[at + (int64_t)taps * 2 * n_pad * 8] = __builtin_bit_cast(unsigned short, h2);
    } else {
      dst[at] = cvt16(v, dt);
    }
  }
}

__global__ void weight_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int cout_w, int cin_w,
                                   int ksize, int kind, int dt, int n_pad, int n_off) {
  const bool phased = kind == 1 || kind == 4;
  const int taps = (kind == 1 || kind == 2 || kind == 4 || kind == 5) ? 4 : ksize * ksize;
  const int kdim = kind == 2 ? 4 * cin_w : (kind == 5 ? 4 * cout_w : ((kind == 3 || kind == 4) ? cout_w : cin_w));
  const int ndim = (kind == 3 || kind == 4 || kind == 5) ? cin_w : cout_w;
  const int64_t total = (phased ? 4 : 1) * (int64_t)(kdim / 16) * taps * 2 * ndim * 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    weight_pack_elem(w, dst, cout_w, cin_w, ksize, kind, dt, n_pad, n_off, i);
}

// Kinds 0 and 3 of a 3x3 weight, one thread per (k, n) pair: it reads the pair's nine taps -- 36 contiguous bytes; the
// eight k (kind 0) or eight n (kind 3) neighbours of a wave make 288-byte runs, every fetched line is used whole -- and
// writes nine (x pieces) 16-bit values, each of them one of 64 consecutive ones of the wave (128-byte stores).  The
// one-thread-per-output-value kernel above fetched every line nine times, 4 bytes of 36 at a time (weights are re-packed
// after every optimizer step: 138 launches per training step).
__device__ __forceinline__ void weight_pack3x3_pair(const float* __restrict__ w, unsigned short* __restrict__ dst, int cout_w,
                                                    int cin_w, int kind, int dt, int n_pad, int n_off, int64_t i) {
  const int ndim = kind == 3 ? cin_w : cout_w;
  const int np = dt == 0 ? 2 : 1;
  {
    // (32-bit index arithmetic: an item index is below cout x cin < 2^31, and a 64-bit division is ~100 instructions -- the batched
    //  refresh of all weights was bound by them)
    const unsigned u = (unsigned)i;
    const int j = (int)(u & 7u);
    unsigned r = u >> 3;
    const unsigned rq = r / (unsigned)ndim;
    const int nn = (int)(r - rq * (unsigned)ndim);
    const int g = (int)(rq & 1u);
    const int q = (int)(rq >> 1);
    const int kk = q * 16 + g * 8 + j;
    const float* wp = w + (kind == 0 ? ((int64_t)nn * cin_w + kk) : ((int64_t)kk * cin_w + nn)) * 9;
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = wp[t];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float x = v[kind == 0 ? tap : 8 - tap];
      const int64_t at = (((((int64_t)q * np + 0) * 9 + tap) * 2 + g) * n_pad + n_off + nn) * 8 + j;
      if (dt == 0) {
        const _Float16 h1 = (_Float16)x;
        const _Float16 h2 = (_Float16)((x - (float)h1) * 2048.0f);
        dst[at] = __builtin_bit_cast(unsigned short, h1);
        dst[at + (int64_t)9 * 2 * n_pad * 8] = __builtin_bit_cast(unsigned short, h2);
      } else {
        dst[at] = cvt16(x, dt);
      }
    }
  }
}

__global__ void weight_pack3x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int cout_w, int cin_w,
                                      int kind, int dt, int n_pad, int n_off) {
  const int kdim = kind == 3 ? cout_w : cin_w, ndim = kind == 3 ? cin_w : cout_w;
  const int64_t total = (int64_t)(kdim / 16) * 2 * ndim * 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    weight_pack3x3_pair(w, dst, cout_w, cin_w, kind, dt, n_pad, n_off, i);
}

// One launch for a whole table of weight-refresh jobs (dsg_conv_weight_pack_batch): a training step re-packs every conv
// weight after the optimizer step -- 140 launches of ~5 us of work each in the mixed-precision tape, plus 44 device copies
// of the time-embedding projection rows.  Job j owns the work items [first[j], first[j + 1]); a thread finds its job by
// bisection and runs the body of the kernel that would have served it (same arithmetic, same bits).
__global__ __launch_bounds__(256) void weight_pack_batch_kernel(const dsg_pack_job* __restrict__ jobs, const int64_t* __restrict__ first,
                                                                int njobs) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= first[njobs]) return;
  int lo = 0, hi = njobs;  // first[lo] <= i < first[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (first[mid] <= i) lo = mid;
    else hi = mid;
  }
  const dsg_pack_job jb = jobs[lo];
  const int64_t k = i - first[lo];
  if (jb.kind < 0) {  // plain copy of fp32 elements
    static_cast<float*>(jb.dst)[k] = jb.w[k];
  } else if (jb.ksize == 3 && (jb.kind == 0 || jb.kind == 3)) {
    weight_pack3x3_pair(jb.w, static_cast<unsigned short*>(jb.dst), jb.cout, jb.cin, jb.kind, jb.dtype, jb.n_pad, jb.n_off, k);
  } else {
    weight_pack_elem(jb.w, static_cast<unsigned short*>(jb.dst), jb.cout, jb.cin, jb.ksize, jb.kind, jb.dtype, jb.n_pad, jb.n_off, k);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Pre-staged operand image of a conv source (dsg_conv_operand_prepare; read by conv_h2_kernel's PRE form).
// One thread per (image, channel block, PADDED pixel): the pixel's 8 channels -> GroupNorm affine + SiLU, or the range
// guard's power-of-two pre-scale for a raw source -> the (hi, 2^11-scaled lo) fp16 pair, exactly the arithmetic of the
// conv's own staging pass (conv_h2_kernel.h: to_operand), so a conv gives the same bits either way.  Border pixels are
// zeros: the conv's zero padding.  grid = (ceil((h+2)(w+2) / 256), (c0 + c1) / 8, n).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_operand_kernel(const float* __restrict__ src0, int c0, const float* __restrict__ src1,
                                                           int c1, int hin, int win, const float* __restrict__ ss, int silu,
                                                           const unsigned* __restrict__ bound0,
                                                           const unsigned* __restrict__ bound1, unsigned short* __restrict__ dst,
                                                           size_t piece_halfs) {
  const int cb = blockIdx.y, n = blockIdx.z, c = c0 + c1;
  const int wp = win + 2, ppos = (hin + 2) * wp;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= ppos) return;
  const int py = pos / wp, px = pos - py * wp;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
  if (py >= 1 && py <= hin && px >= 1 && px <= win) {
    const float* sp = (cb * 8 < c0) ? src0 + (((size_t)n * (c0 / 8) + cb) * hin * win) * 8
                                    : src1 + (((size_t)n * (c1 / 8) + (cb - c0 / 8)) * hin * win) * 8;
    const float4* q = reinterpret_cast<const float4*>(sp + ((size_t)(py - 1) * win + (px - 1)) * 8);
    const float4 v0 = q[0], v1 = q[1];
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    float xs = 1.f;
    if (ss == nullptr && bound0 != nullptr) {  // (the conv's own guard: conv_h2_kernel.h, "range guard")
      unsigned b = bound0[n];
      if (bound1 != nullptr) b = max(b, bound1[n]);
      const int e = min(100, max(-100, (int)(b >> 23) - 127));
      if (b != 0u && (e > 12 || e < -6)) xs = __uint_as_float((unsigned)(127 - e) << 23);
    }
    unsigned w1[4], w2[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      float a = v[2 * jp], b = v[2 * jp + 1];
      if (ss != nullptr) {
        const float* e = ss + ((size_t)n * c + cb * 8 + 2 * jp) * 2;  // (scale, shift) of the pair's two channels
        a = a * e[0] + e[1];
        b = b * e[2] + e[3];
      } else {
        a *= xs;
        b *= xs;
      }
      // silu(v) = v * r, r = 1 / (1 + exp(-v)).  The conv kernel's staging pass (ACT = 2: the activation is a compile-time
      // fact there) takes the low-order piece from the EXACT product, fma(v, r, -hi) -- hipcc contracts its
      // `v * r - hi` -- and so does this kernel, explicitly: same bits (tests/test_gpu_operand.py holds both to it).
      float ra = 1.f, rb = 1.f;
      if (silu) {
        ra = __builtin_amdgcn_rcpf(1.0f + __expf(-a));
        rb = __builtin_amdgcn_rcpf(1.0f + __expf(-b));
      }
      const _Float16 a1 = (_Float16)(a * ra), b1 = (_Float16)(b * rb);
      const half2v h = {a1, b1};
      const half2v l = {(_Float16)(__builtin_fmaf(a, ra, -(float)a1) * 2048.0f), (_Float16)(__builtin_fmaf(b, rb, -(float)b1) * 2048.0f)};
      w1[jp] = __builtin_bit_cast(unsigned, h);
      w2[jp] = __builtin_bit_cast(unsigned, l);
    }
    hi = u32x4{w1[0], w1[1], w1[2], w1[3]};
    lo = u32x4{w2[0], w2[1], w2[2], w2[3]};
  }
  unsigned short* o = dst + (((size_t)n * (c / 8) + cb) * ppos + pos) * 8;
  *reinterpret_cast<u32x4*>(o) = hi;
  *reinterpret_cast<u32x4*>(o + piece_halfs) = lo;
}

}  // namespace dsg

DSG_API int dsg_conv_operand_bytes(int32_t n, int32_t c, int32_t hin, int32_t win, int32_t dtype, size_t* bytes) {
  DSG_CHECK_ARG(bytes != nullptr, "dsg_conv_operand_bytes: bytes is NULL");
  DSG_CHECK_ARG(n > 0 && c > 0 && c % 16 == 0 && hin > 0 && win > 0, "dsg_conv_operand_bytes: bad dims (c %% 16 != 0?)");
  DSG_CHECK_SHAPE(dtype == DSG_F32, "dsg_conv_operand_bytes: operand images exist for the fp32-equivalent mode only (dtype %d)", dtype);
  *bytes = 2 * (size_t)n * c * (hin + 2) * (win + 2) * 2;
  return DSG_OK;
}

DSG_API int dsg_conv_operand_prepare(const float* src0, int32_t c0, const float* src1, int32_t c1, int32_t n, int32_t hin,
                                     int32_t win, const float* gn_scale_shift, int32_t silu, const uint32_t* src_bound,
                                     const uint32_t* src_bound1, void* operand, int32_t dtype, void* stream) {
  DSG_CHECK_ARG(src0 && operand, "dsg_conv_operand_prepare: NULL pointer");
  DSG_CHECK_ARG((c1 == 0) == (src1 == nullptr), "dsg_conv_operand_prepare: src1/c1 mismatch");
  DSG_CHECK_ARG(n > 0 && n <= 65535 && c0 > 0 && c1 >= 0 && c0 % 8 == 0 && c1 % 8 == 0 && (c0 + c1) % 16 == 0 && hin > 0 && win > 0,
                "dsg_conv_operand_prepare: bad dims (channel-blocked sources: c0 %% 8, c1 %% 8, (c0 + c1) %% 16)");
  DSG_CHECK_ARG(!(silu && !gn_scale_shift), "dsg_conv_operand_prepare: silu without gn_scale_shift (no conv of the U-Net has that)");
  DSG_CHECK_SHAPE(dtype == DSG_F32, "dsg_conv_operand_prepare: operand images exist for the fp32-equivalent mode only (dtype %d)", dtype);
  const int ppos = (hin + 2) * (win + 2);
  hipLaunchKernelGGL(dsg::conv_operand_kernel, dim3((ppos + 255) / 256, (c0 + c1) / 8, n), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src0, c0, src1, c1, hin, win, gn_scale_shift, silu,
                     gn_scale_shift ? nullptr : src_bound, gn_scale_shift ? nullptr : src_bound1,
                     static_cast<unsigned short*>(operand), (size_t)n * (c0 + c1) * ppos);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

static int pack_dims(int32_t cout, int32_t cin, int32_t ksize, int32_t kind, int32_t dtype, int32_t n_total,
                     int* kdim, int* ndim, int* taps, int* phases, int* n_pad) {
  DSG_CHECK_ARG(kind >= 0 && kind <= 5, "dsg_conv_weight_pack: kind must be 0..5 (got %d)", kind);
  DSG_CHECK_ARG(dtype >= DSG_F32 && dtype <= DSG_F16, "dsg_conv_weight_pack: dtype must be DSG_F32 / DSG_BF16 / DSG_F16");
  DSG_CHECK_ARG(cout > 0 && cin > 0 && (ksize == 1 || ksize == 3), "dsg_conv_weight_pack: bad dims");
  DSG_CHECK_ARG(kind == 0 || kind == 3 || ksize == 3, "dsg_conv_weight_pack: kinds 1, 2, 4, 5 are 3x3 only");
  DSG_CHECK_ARG(kind != 2 || cin % 8 == 0, "dsg_conv_weight_pack: the stride-2 form needs cin %% 8 == 0 (got %d)", cin);
  DSG_CHECK_ARG(kind != 5 || cout % 8 == 0, "dsg_conv_weight_pack: the up-sampler's data-gradient form needs cout %% 8 == 0 (got %d)", cout);
  *kdim = kind == 2 ? 4 * cin : (kind == 5 ? 4 * cout : ((kind == 3 || kind == 4) ? cout : cin));
  *ndim = (kind == 3 || kind == 4 || kind == 5) ? cin : cout;
  *taps = (kind == 1 || kind == 2 || kind == 4 || kind == 5) ? 4 : ksize * ksize;
  *phases = (kind == 1 || kind == 4) ? 4 : 1;
  DSG_CHECK_ARG(*kdim % 16 == 0, "dsg_conv_weight_pack: the contraction axis (%d) must be a multiple of 16", *kdim);
  if (n_total == 0) n_total = *ndim;
  *n_pad = (n_total + 63) / 64 * 64;
  return DSG_OK;
}

DSG_API int dsg_conv_weight_pack_bytes(int32_t cout, int32_t cin, int32_t ksize, int32_t kind, int32_t dtype,
                                       int32_t n_total, size_t* bytes) {
  DSG_CHECK_ARG(bytes != nullptr, "dsg_conv_weight_pack_bytes: bytes is NULL");
  int kdim, ndim, taps, phases, n_pad;
  const int rc = pack_dims(cout, cin, ksize, kind, dtype, n_total, &kdim, &ndim, &taps, &phases, &n_pad);
  if (rc != DSG_OK) return rc;
  *bytes = (size_t)phases * (kdim / 16) * (dtype == DSG_F32 ? 2 : 1) * taps * 2 * n_pad * 8 * 2;
  return DSG_OK;
}

DSG_API int dsg_conv_weight_pack(const float* w_oihw, void* dst, int32_t cout, int32_t cin, int32_t ksize, int32_t kind,
                                 int32_t dtype, int32_t n_total, int32_t n_off, void* stream) {
  DSG_CHECK_ARG(w_oihw && dst, "dsg_conv_weight_pack: NULL pointer");
  int kdim, ndim, taps, phases, n_pad;
  const int rc = pack_dims(cout, cin, ksize, kind, dtype, n_total, &kdim, &ndim, &taps, &phases, &n_pad);
  if (rc != DSG_OK) return rc;
  DSG_CHECK_ARG(n_off >= 0 && n_off + ndim <= (n_total ? n_total : ndim), "dsg_conv_weight_pack: column window out of range");
  DSG_CHECK_ARG((kind == 0 || kind == 3) || (n_total == 0 && n_off == 0), "dsg_conv_weight_pack: column windows are for kinds 0 and 3");
  if (ksize == 3 && (kind == 0 || kind == 3)) {
    const int64_t pairs = (int64_t)kdim * ndim;
    hipLaunchKernelGGL(dsg::weight_pack3x3_kernel, dim3((unsigned)std::min<int64_t>(dsg::cdiv64(pairs, 256), 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w_oihw, static_cast<unsigned short*>(dst), cout, cin, kind, dtype, n_pad,
                       n_off);
    DSG_LAUNCH_CHECK();
    return DSG_OK;
  }
  const int64_t total = (int64_t)phases * (kdim / 16) * taps * 2 * ndim * 8;
  const int blocks = (int)std::min<int64_t>(dsg::cdiv64(total, 256), 4096);
  hipLaunchKernelGGL(dsg::weight_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw,
                     static_cast<unsigned short*>(dst), cout, cin, ksize, kind, dtype, n_pad, n_off);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

// work items of one job (threads of weight_pack_batch_kernel): pairs for the 3x3 forward / data-gradient forms, output
// elements otherwise, fp32 elements for a copy
DSG_API int dsg_conv_weight_pack_batch_items(const dsg_pack_job* job, int64_t* items) {
  DSG_CHECK_ARG(job != nullptr && items != nullptr, "dsg_conv_weight_pack_batch_items: NULL pointer");
  if (job->kind < 0) {
    DSG_CHECK_ARG(job->cout > 0, "dsg_conv_weight_pack_batch_items: a copy job needs its element count in `cout`");
    *items = job->cout;
    return DSG_OK;
  }
  int kdim, ndim, taps, phases, n_pad;
  const int rc = pack_dims(job->cout, job->cin, job->ksize, job->kind, job->dtype, job->n_total, &kdim, &ndim, &taps, &phases, &n_pad);
  if (rc != DSG_OK) return rc;
  DSG_CHECK_ARG(job->n_pad == n_pad, "dsg_conv_weight_pack_batch_items: n_pad %d != %d (cout padded to 64 of the packed matrix)", job->n_pad, n_pad);
  if (job->ksize == 3 && (job->kind == 0 || job->kind == 3)) *items = (int64_t)(kdim / 16) * 2 * ndim * 8;
  else *items = (int64_t)phases * (kdim / 16) * taps * 2 * ndim * 8;
  return DSG_OK;
}

DSG_API int dsg_conv_weight_pack_batch(const dsg_pack_job* jobs_dev, const int64_t* first_dev, int32_t njobs, int64_t total_items,
                                       void* stream) {
  DSG_CHECK_ARG(jobs_dev && first_dev && njobs > 0 && total_items > 0, "dsg_conv_weight_pack_batch: bad argument");
  DSG_CHECK_ARG(total_items <= (int64_t)256 * 0x7FFFFFFF, "dsg_conv_weight_pack_batch: too many work items");
  hipLaunchKernelGGL(dsg::weight_pack_batch_kernel, dim3((unsigned)dsg::cdiv64(total_items, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), jobs_dev, first_dev, njobs);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

// the fp32-equivalent (split) forms under their round-1 names
DSG_API int dsg_conv_weight_relayout_h2(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, int32_t ksize,
                                        int32_t cout_total, int32_t cout_off, void* stream) {
  return dsg_conv_weight_pack(w_oihw, dst_half, cout, cin, ksize, 0, DSG_F32, cout_total, cout_off, stream);
}
DSG_API int dsg_conv_weight_relayout_h2_fold(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, void* stream) {
  return dsg_conv_weight_pack(w_oihw, dst_half, cout, cin, 3, 1, DSG_F32, 0, 0, stream);
}
DSG_API int dsg_conv_weight_relayout_h2_s2(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, void* stream) {
  DSG_CHECK_ARG(cin > 0 && cin % 8 == 0, "dsg_conv_weight_relayout_h2_s2: cin (%d) must be a positive multiple of 8", cin);
  return dsg_conv_weight_pack(w_oihw, dst_half, cout, cin, 3, 2, DSG_F32, 0, 0, stream);
}
DSG_API int dsg_conv_weight_relayout_h2_dgrad(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin,
                                              int32_t ksize, void* stream) {
  return dsg_conv_weight_pack(w_oihw, dst_half, cout, cin, ksize, 3, DSG_F32, 0, 0, stream);
}
