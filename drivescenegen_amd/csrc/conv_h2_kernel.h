// 3x3 stride-1 convolution with fp32-equivalent accuracy on the fp16 matrix cores ("fp16x2 split").
//
// Same contract, tiling and fusions as conv.hip's conv_mfma_kernel (reference call sites: every 3x3
// Conv2d of diffusers' UNet2DModel as built at DriveSceneGen/scripts/train.py:39-57 and run at
// DriveSceneGen/pipeline/training_pipeline.py:84), but the contraction runs at the 16x-faster f16 MFMA
// rate without giving up fp32 accuracy:
//
//   x = x1 + x2 * 2^-11,  x1 = fp16(x),  x2 = fp16((x - x1) * 2^11)      (|x - x1 - x2*2^-11| <= 2^-24 |x|)
//   w = w1 + w2 * 2^-11   likewise (split once, at weight re-layout time)
//   sum w*x  ~=  sum w1*x1  +  2^-11 * sum (w1*x2 + w2*x1)              (dropped w2*x2 term: 2^-24 relative)
//
// i.e. 3 v_mfma_f32_32x32x16_f16 per 16-deep k-step instead of 8 v_mfma_f32_32x32x2_f32: 5.3x fewer matrix
// cycles.  fp16 x fp16 products are exact in the fp32 accumulator; the scaled low-order products go to a
// second accumulator so that nothing is lost to fp16's narrow exponent (the 2^11 pre-scale keeps the low
// parts normal).  Measured error vs fp64 is at or below that of a sequential fp32 fmaf chain
// (tests/test_gpu_ops.py::test_conv_h2_*).  Operands outside fp16's range are handled by the range guard: sources
// without a norm in front are pre-scaled by the power of two of their per-image bound (ConvH2P::bound0 / bound1),
// weights outside [2^-8, 3e4] send the conv to the exact f32-MFMA kernel (unet.hip: Conv::off_split).
//
// LDS images (per K-chunk of 16 channels, double-buffered; same bytes as the fp32 kernel's):
//   X[piece 2][g 2][pos 10x34][8 halfs]   -- lane = pixel reads one 16-B fragment (k-group g = lane>>5)
//   W[piece 2][tap 9][g 2][cout 64][8]    -- lane = cout  reads one 16-B fragment; filled by LDS-DMA
//                                            (global_load_lds_dwordx4: the pre-split weights need no math)
// A and B use the same (g, j) <-> channel 8g+j map, so the MFMA's internal k order is irrelevant.
#pragma once
#include "dsg_h16.h"
#include <algorithm>
#include <type_traits>

namespace dsg {

bool prof_on();
int prof_begin(int kid, double flops, double bytes, hipStream_t st);
void prof_end(int idx, hipStream_t st);

struct ConvH2P {
  const void* src0;  // fp32, or the 16-bit type of PREC when channel-blocked (see dsg_h16.h)
  const void* src1;
  int c0, c1, cin;
  int n, hin, win;
  int hc, wc;
  int hout, wout;
  int cout, cout_pad;
  int wh_stride;       // couts per weight row (>= cout_pad when `wh` is a column window of a wider matrix)
  const void* wh;      // [cin/16][pieces][9][2][wh_stride][8] 16-bit values; pieces = 2 (hi, scaled lo) for PREC 0, else 1
  const float* bias;
  const float* ss;
  int silu;
  const float* temb;
  int temb_stride;
  const void* res;
  void* dst;
  double* stats;  // optional [n][cout][hout/8 * wout/32][2]: per-tile (sum, sum of squares) of the values written
  int tiles_x, tiles_y;
  // Range guard of the fp16x2 split (PREC 0): per-image upper bounds of max|x| as float bits (positive floats order
  // like their bits).  bound0 / bound1 describe the two sources of a call WITHOUT a norm in front (shortcuts, up- /
  // down-sampler convs): when the bound leaves [2^-6, 2^12] the patch is multiplied by the power of two that brings it
  // to ~1 before the split and the accumulators by its inverse -- exact, so |x| = 1e5 neither overflows the fp16 pieces
  // nor changes the result.  (The bounds come from the GroupNorm statistics: dsg_gn_finalize_parts_bound,
  // dsg_range_bound_from_stats.)
  const unsigned* bound0;
  const unsigned* bound1;
  // Split-K (small batches: grids smaller than the chip): gridDim.y workgroups share a tile, each contracts a
  // contiguous run of K-chunks and writes its fp32 partial sums to slab blockIdx.y of `dst` (split_stride bytes
  // apart; bias / temb / residual / statistics are then the reduce pass's: splitk_reduce_blk_kernel)
  size_t split_stride;
  // Fused shortcut (SC kernels): the resnet's 1x1 conv_shortcut over its UN-normalised input (diffusers ResnetBlock2D:
  // output = conv_shortcut(x) + conv2(...), the blocks train.py:39-57 builds wherever in != out channels) rides on this
  // conv2's accumulators as sc_cin / 16 extra K-chunks of ONE tap each, read through the same halo-patch addressing:
  // the shortcut's result is never written to HBM and never read back as a residual.  bound0 / bound1 are then the
  // range-guard bounds of sc_src0 / sc_src1 (the main source is normalised); sc_bias is added with bias / temb.
  const void* sc_src0;
  const void* sc_src1;
  int sc_c0, sc_c1, sc_cin;
  const void* sc_wh;   // [sc_cin/16][pieces][1][2][sc_wh_stride][8]
  int sc_wh_stride;
  const float* sc_bias;
  // Pre-staged operand image (PRE kernels): what the staging pass would have written to LDS -- GroupNorm affine + SiLU
  // (or the range guard's pre-scale) applied, split into the (hi, scaled lo) fp16 pair -- computed ONCE by
  // dsg_conv_operand_prepare instead of once per cout tile: [piece][N][cin/8][hin + 2][win + 2][8] 16-bit values with a
  // zero border, so a tile's halo patch is DMA'd to LDS as it is (no registers, no VALU, no bounds logic).
  const void* pre;
  size_t pre_piece_stride;  // bytes between the two pieces
  // GroupNorm-backward statistics from the data-gradient conv's epilogue (GNB kernels).  The conv writes dA, the gradient
  // w.r.t. the ACTIVATED tensor a = silu(x * sc + sh) that the forward conv read; the norm's backward needs, per (n, c),
  // S1 = sum du and S2 = sum du * xhat with du = dA * silu'(x * sc + sh) -- a pass of its own over x and dA
  // (gn_bwd_stats*_kernel: 6.3 % of the bf16 training step, 3.9 % of the fp32 one).  Here the epilogue, which holds dA in
  // registers, reads the pre-norm x of its tile (layout and shape of dst; the channel tile lies in ONE of the two concatenated
  // sources: the host checks) and the (sc, sh) of its channels, and leaves per-tile partials (sum du, sum du * x) in `stats`
  // -- the forward statistics' table, [n][cout][tiles][2] -- RAW second moment: xhat = (x - mean) * rstd is applied to the sums
  // by gnb_parts_reduce_kernel in fp64.  x is read once more, the statistics pass's two reads disappear.
  const void* gnb_x0;
  const void* gnb_x1;
  int gnb_c0;            // channels of gnb_x0 (the rest, cout - gnb_c0, are gnb_x1's)
  const float* gnb_ss;   // [n][cout][2] (scale, shift) of the norm
  int gnb_silu;
};

constexpr int H2_TW = 32, H2_KC = 16, H2_BM = 64;

// NT = output rows per wave (2 or 4): a workgroup covers 4*NT rows x 32 cols.  NT = 4 halves the LDS operand
// traffic per MFMA (each weight fragment feeds 4 pixel tiles) and the weight DMA per MFMA; it needs 256
// accumulator registers (the kernel owns the SIMD: 1 wave, 512 registers).
// KS = 3 (halo of 1) or 1 (no halo; attention projections and resnet shortcuts)
// NW = waves per workgroup (4: one per SIMD with the whole register file; 8: two per SIMD with half of it each,
// so that one wave's staging / LDS / wait time is covered by the other's MFMAs)
template <int NT, int KS, int NW = 4, int TAPS_ = KS * KS, int BM_ = 64, int NP_ = 2, int PRE_ = 0>
struct H2Geom {
  static constexpr int NP = NP_;                      // operand pieces: 2 (hi + scaled lo, fp16x2 split) or 1 (bf16 / fp16)
  static constexpr int BM = BM_;                      // output channels per workgroup: 64, or 32 (two workgroups per CU)
  static constexpr int NTH = 64 * NW;
  static constexpr int TAPS = TAPS_;                  // 4 in the folded up-sampler mode (2x2 taps of the 3x3 patch)
  static constexpr int TH = NW * NT;
  static constexpr int PH = TH + KS - 1;
  static constexpr int PW = H2_TW + KS - 1;
  static constexpr int PSZ = PW * PH;                 // KS=3: 340 (NT=2) / 612 (NT=4); KS=1: 256 / 512
  static constexpr int PUNITS = (PSZ + 63) / 64;      // PRE: 1-KB DMA units (64 positions x 16 B) per (piece, k-group) region
  static constexpr int PSTR = PRE_ ? PUNITS * 64 : PSZ;  // positions between regions (PRE: whole DMA units, so a unit never spills into the next region)
  static constexpr int WHALFS = NP * TAPS * 2 * BM * 8;  // [piece][tap][g][cout][8]: 36864 B / 4096 B at BM = 64, NP = 2
  static constexpr int XHALFS = NP * 2 * PSTR * 8;    // [piece][g][pos][8]
  static constexpr int BUF_BYTES = (WHALFS + XHALFS) * 2 + 64;  // + a dump slot for masked lanes
  static constexpr int FULL = PSZ / NTH;              // full NTH-position slabs per k-group
  static constexpr bool HAS_REM = (PSZ % NTH) != 0;   // KS=3 leaves a remainder slab shared by the two k-groups
  static constexpr int NU = 2 * FULL + (HAS_REM ? 1 : 0);  // staging units per thread
  static constexpr int REM0 = FULL * NTH;             // first position of the remainder unit
  static constexpr int NSEG = 2 * NP * TAPS;          // (piece, tap, g) weight segments per chunk, BM x 16 bytes each
  static constexpr int NUNIT = NSEG * BM / 64;        // 1-KB DMA units per chunk (a unit = 64 / BM segments)
  static constexpr int NDMA = (NUNIT + NW - 1) / NW;  // weight DMAs per wave per chunk
  static_assert(PSZ - REM0 <= NTH / 2, "the remainder slab must fit half the workgroup per k-group");
};

// x + (x of the lane selected by a DPP control): the building block of a fixed-order 32-lane tree sum
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float x) {
  const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, true);
  return x + __int_as_float(y);
}
// after this, lanes 16..31 hold the sum over lanes 0..31 and lanes 48..63 the sum over lanes 32..63
__device__ __forceinline__ float half_wave_sum(float x) {
  x = dpp_add<0xB1>(x);        // quad_perm [1,0,3,2]
  x = dpp_add<0x4E>(x);        // quad_perm [2,3,0,1]
  x = dpp_add<0x141>(x);       // row_half_mirror
  x = dpp_add<0x140>(x);       // row_mirror: every lane of a 16-row holds the row sum
  x = dpp_add<0x142, 0xA>(x);  // row_bcast15 into rows 1 and 3
  return x;
}

// a uniform 64-bit address as an SGPR pair, whatever register class the compiler had chosen for it (with the scalar file
// full it keeps uniform values in VGPRs, which an "s" asm operand cannot take)
__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ float silu_fast_h(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// d/du (u * sigmoid(u)), the arithmetic of train_ops.hip's dsilu (hardware exp / reciprocal): the GNB epilogue's statistics are in
// the rounding class of the standalone statistics pass
__device__ __forceinline__ float dsilu_fast(float u) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
  return s * (1.0f + u * (1.0f - s));
}

// GM: 3 STRIDE-2 3x3 conv (Downsample2D) as a 2x2 conv over the space-to-depth image, which for channel-blocked
// sources is pure addressing: k-group (cb, py, px) of the 4C "channels" is channel block cb read at pixels
// (2y + py, 2x + px); output row oy needs input rows 2oy-1 (y' = oy-1, py = 1), 2oy (oy, 0), 2oy+1 (oy, 1), i.e. the
// 2x2 corner {y'-1, y'} x {x'-1, x'} of the low-resolution patch with zero weights where no 3x3 tap lands (7 of the
// 16 (tap, phase) pairs): 16/9 of the useful products, on the pipe that is 5x faster than the f32 one.
// GM: 0 plain, 1 nearest x2 gather, 2 nearest x2 FOLDED into the weights: Upsample2D + 3x3 conv is four 2x2 convs of
// the low-resolution input, one per output phase (py, px) = (Y & 1, X & 1): rows {y-1: W0, y: W1+W2} for py = 0 and
// {y: W0+W1, y+1: W2} for py = 1, likewise in x -- 16 tap products per input pixel instead of 36.  The kernel runs on
// the low-resolution grid with the phase as an extra (outer) cout-tile index, walks the phase's 2x2 corner of the 3x3
// patch and scatters its results to the (2y+py, 2x+px) pixels.
// NT: rows per wave; KS: 3 | 1.
// Staging units are arranged so that the k-group g (hence the channel plane and the GroupNorm scale/shift) of
// every unit is WAVE-UNIFORM: channel-plane bases and scale/shift live in SGPRs (s_load / saddr-form global
// loads), and the only per-lane address is the 32-bit halo offset computed once per tile.
//   units 0..FULL-1: g = 0, halo positions tid + 256*i      units FULL..2*FULL-1: g = 1, same positions
//   last unit: g = wave >> 1, halo position FULL*256 + (tid & 127)   (the remainder, valid where < PSZ)
// ACT: 0 the input is used as it is; 2 GroupNorm affine + SiLU; 3 decided at run time from p.ss / p.silu
// OCC: workgroups the kernel is compiled to fit per CU (register budget 512 / (OCC * NW / 4) per lane)
// LAY: bit 0: the sources are channel-blocked [N][C/8][H][W][8] (a halo position's k-group is 32 contiguous bytes:
//      two 16-byte loads instead of eight dword gathers from eight channel planes); bit 1: dst / residual are
//      (a lane's four consecutive output channels are one 16-byte store; a wave instruction writes 1 KB contiguous)
// BM: output channels per workgroup.  128 (16-bit modes): four 32-channel MFMA tiles share one staged patch -- a third of
//     the MFMAs per K-chunk but the same GroupNorm / SiLU / rounding work per patch makes the 64-cout geometry
//     staging-bound there; with 128 couts the accumulators fill the register file as the split's two sets do.
//     32 (with NT = 2, OCC = 2) is the small-workgroup geometry for the shallow levels:
//     80 KB of LDS and half the register file, so two workgroups share a CU and one's patch loads and output stores
//     run under the other's MFMAs (a workgroup that owns the CU runs those phases back to back).
// PREC: 0 fp32 tensors, fp16x2-split products; 1 bf16 / 2 fp16: channel-blocked tensors are 16-bit in HBM, one MFMA per
//       product, fp32 accumulate, GroupNorm affine + SiLU in fp32 before the operand is rounded ([N,C,H,W] tensors stay fp32)
// WS: ONE weight slab in LDS instead of two (the patch stays double-buffered): 80 KB per workgroup at 64 couts x 8 rows
//     with the split's two-piece operands, i.e. two workgroups per CU for the 64- / 128-channel levels, whose tiles are
//     short (4-8 K-chunks) and spend a third of their life in the prologue and epilogue.  The slab of chunk q+1 can only
//     be fetched once every wave is done with chunk q -- that DMA latency is exposed per chunk and, like the prologue and
//     the epilogue, covered by the CU's other workgroup.
// SC: 1 = the fused-shortcut form (see ConvH2P::sc_*): after the nq 3x3 chunks of the normalised source come
//     sc_cin / 16 one-tap chunks of the raw shortcut source(s), staged through the same patch slots (centre tap only)
// PRE: 1 = the main source is a pre-staged operand image (ConvH2P::pre; ACT must be 0: the activation is in the image).
//      The K loop then stages nothing: a chunk's halo patch -- NP * 2 regions of PSZ positions x 16 B -- goes global -> LDS
//      in 1-KB DMA units like the weights (lane L of a unit fetches halo position 64 u + L: its address inside a padded
//      channel block is the only per-lane quantity, computed once per tile), so the loop is MFMAs + fragment reads + ~20
//      DMA issues per chunk.  For the deep levels, where cout / 64 workgroups would each re-normalise, re-activate and
//      re-split the same patch (the loop measured 1020 cycles per tap against 755 with nothing staged).
template <int GM, int NT, int KS, int ACT = 3, int NW = 4, int OCC = 1, int LAY = 0, int BM = 64, int PREC = 0, int WS = 0, int SC = 0, int PRE = 0, int GNB = 0>
__global__ __launch_bounds__(64 * NW, OCC * NW / 4) void conv_h2_kernel(ConvH2P p) {
  static_assert(!GNB || (KS == 3 && GM == 0 && ACT == 0 && !SC && !PRE && !WS),
                "GroupNorm-backward statistics: the plain stride-1 3x3 data-gradient conv");
  static_assert(!WS || KS == 3, "the one-slab layout is for the 3x3 kernels");
  static_assert(!SC || (KS == 3 && GM == 0 && (ACT == 2 || PRE) && LAY == 3),
                "fused shortcut: plain 3x3 conv2 of a resnet, channel-blocked tensors");
  static_assert(!PRE || (KS == 3 && (GM == 0 || GM == 2) && ACT == 0 && LAY == 3 && !WS),
                "pre-staged operands: stride-1 3x3 convs (plain or folded up-sampler) on channel-blocked tensors");
  using K0 = std::integral_constant<int, 0>;   // operand kinds: 0 = the main source (GroupNorm affine + SiLU as configured),
  using K1 = std::integral_constant<int, 1>;   //                1 = the fused shortcut's raw source
  constexpr bool SB = (LAY & 1) != 0, DB = (LAY & 2) != 0;
  constexpr int NP = PREC ? 1 : 2;              // operand pieces
  constexpr bool S16 = PREC != 0 && SB;         // 16-bit sources (8 channels of a pixel = one 16-byte load)
  constexpr int ESS = S16 ? 2 : 4;              // bytes per source element
  constexpr int ESD = (PREC != 0 && DB) ? 2 : 4;  // bytes per dst / residual element
  constexpr int MTN = BM / 32;  // 32-channel MFMA tiles per workgroup
  constexpr bool S2D = GM == 3 || GM == 4;    // 2x2 taps over the space-to-depth image (GM 4: the 4x4 stride-2 window, see load_frags)
  using G = H2Geom<NT, KS, NW, (GM == 2 || S2D) ? 4 : KS * KS, BM, NP, PRE>;
  constexpr int NTH = G::NTH;
  constexpr int H2_TH = G::TH, H2_PSZ = G::PSZ, H2_XHALFS = G::XHALFS, H2_BUF_BYTES = G::BUF_BYTES, H2_NU = G::NU;
  constexpr int H2_PSTR = G::PSTR;  // positions between the (piece, k-group) regions of a patch
  constexpr int FULL = G::FULL, TAPS = G::TAPS, H2_PW = G::PW, H2_WHALFS = G::WHALFS, PADK = KS / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
#ifdef DSG_H2_TIMING
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;

  // Workgroup id -> (spatial tile, cout tile).  Consecutive ids go to the 8 XCDs in turn, each with its own L2.
  // XCD k gets a CONTIGUOUS eighth of the spatial tiles, walked in raster order with the cout tiles of one patch
  // on neighbouring ids: workgroups that share input -- the same patch for another cout tile, or the 128-byte
  // lines and halo rows a patch has in common with its left/right/upper/lower neighbours -- run at the same time
  // behind the same L2, so that data comes from HBM once instead of once per XCD.
  const int nct = (p.cout_pad / BM) * (GM == 2 ? 4 : 1), nsp = p.tiles_x * p.tiles_y * p.n;
  int bid, ct;
  if ((nsp & 7) == 0) {
    const int grp = blockIdx.x >> 3;
    ct = grp % nct;
    bid = (blockIdx.x & 7) * (nsp >> 3) + grp / nct;
  } else {
    ct = blockIdx.x % nct;
    bid = blockIdx.x / nct;
  }
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int phase = GM == 2 ? ct / (p.cout_pad / BM) : 0;  // (py, px) = (phase >> 1, phase & 1)
  if (GM == 2) ct -= phase * (p.cout_pad / BM);
  const int m0 = ct * BM;
  const int oy0 = ty * H2_TH, ox0 = tx * H2_TW;
  const int plane = p.hin * p.win;
  int nq = p.cin / H2_KC, qb = 0;  // this workgroup's K-chunks: [qb, qb + nq)
  if (gridDim.y > 1) {
    const int per = (nq + (int)gridDim.y - 1) / (int)gridDim.y;
    qb = (int)blockIdx.y * per;
    nq = min(per, nq - qb);
  }
  const int g2 = wave / (NW / 2);  // k-group of the remainder unit (uniform per wave)

  // Per staging unit: global halo offset, and the LDS slots of its pieces.  Positions outside the image (zero
  // padding) or past the patch write to a dump slot instead; the real slots of padding positions are zeroed once.
  int goff[H2_NU], xoff[H2_NU], xoff2[H2_NU], zoff[H2_NU];
#pragma unroll
  for (int i = 0; i < H2_NU; ++i) {
    const int g = i < FULL ? 0 : (i < 2 * FULL ? 1 : g2);
    const int pos = i < 2 * FULL ? tid + NTH * (i % FULL) : G::REM0 + (tid & (NTH / 2 - 1));
    int off = 0, xo = H2_WHALFS + H2_XHALFS, xo2 = H2_WHALFS + H2_XHALFS, zo = -1;
    if (pos < H2_PSZ) {
      const int py = pos / H2_PW, px = pos - py * H2_PW;
      const int gy = oy0 - PADK + py, gx = ox0 - PADK + px;
      const int slot = H2_WHALFS + (g * H2_PSTR + pos) * 8;  // piece 0; piece 1 is 2*PSTR*8 halfs further
      if (gy >= 0 && gy < p.hc && gx >= 0 && gx < p.wc) {
        off = S2D ? (2 * gy) * p.win + 2 * gx : (GM == 1 ? (gy >> 1) : gy) * p.win + (GM == 1 ? (gx >> 1) : gx);
        xo = slot;
        xo2 = NP == 2 ? slot + 2 * H2_PSTR * 8 : slot;
      } else {
        zo = slot;
      }
    }
    goff[i] = off;
    xoff[i] = xo;
    xoff2[i] = xo2;
    zoff[i] = zo;
  }
  const bool has_ss = ACT == 3 ? p.ss != nullptr : ACT != 0;
  const bool do_silu = ACT == 3 ? (has_ss && p.silu) : ACT == 2;
  const float* ssg = has_ss ? p.ss + (size_t)n * p.cin * 2 : nullptr;
  // range guard (see ConvH2P): xs = 2^-e pre-scale of an un-normalised patch, rg_out = 2^e on the way out; both 1
  // (and the products bit-identical to the unguarded kernel) while the bound is inside the safe range
  float xs = 1.f, rg_out = 1.f;
  if constexpr (PREC == 0 && (ACT != 2 || SC)) {
    if (p.bound0 != nullptr && (SC || !has_ss)) {
      unsigned b = p.bound0[n];
      if (p.bound1 != nullptr) b = max(b, p.bound1[n]);
      b = __builtin_amdgcn_readfirstlane(b);
      const int e = min(100, max(-100, (int)(b >> 23) - 127));  // floor(log2(bound))
      if (b != 0u && (e > 12 || e < -6)) {
        xs = __uint_as_float((unsigned)(127 - e) << 23);
        rg_out = __uint_as_float((unsigned)(127 + e) << 23);
      }
    }
  }

  // raw patch values of one K-chunk: fp32 (8 registers per unit), or the 8 16-bit channels of a pixel as loaded (4)
  struct Patch {
    float f[S16 ? 1 : H2_NU][8];
    unsigned q[S16 ? H2_NU : 1][4];
  };
  Patch xr;
  // GroupNorm (scale, shift) of this image's channels: copied once into LDS behind the two K-chunk buffers; a commit
  // reads its 8 channels from there (uniform address: a broadcast read) instead of carrying them in registers
  // LDS: [W | X | dump] x 2, or with WS [W | X | dump | X | dump]: buffer 1 starts one X region further, so that the
  // same (buffer base + offset) addressing reaches its patch; weights are always read from / fetched into buffer 0's
  constexpr int H2_BUF1_OFF = WS ? H2_XHALFS * 2 + 64 : H2_BUF_BYTES;
  constexpr int H2_LDS_BUFS = WS ? H2_BUF_BYTES + H2_BUF1_OFF : 2 * H2_BUF_BYTES;
  float* ssl = reinterpret_cast<float*>(smem_raw + H2_LDS_BUFS);

  const char* src0b = static_cast<const char*>(p.src0);
  const char* src1b = static_cast<const char*>(p.src1);
  auto src_of = [&](int q) -> const char* {  // uniform
    const int cb = (q + qb) * H2_KC;
    return (cb < p.c0) ? src0b + ((size_t)n * p.c0 + cb) * plane * ESS
                       : src1b + ((size_t)n * p.c1 + (cb - p.c0)) * plane * ESS;
  };
  const char* sc0b = static_cast<const char*>(p.sc_src0);
  const char* sc1b = static_cast<const char*>(p.sc_src1);
  auto sc_src_of = [&](int j) -> const char* {  // chunk j of the shortcut's source(s) (uniform)
    const int cb = j * H2_KC;
    return (cb < p.sc_c0) ? sc0b + ((size_t)n * p.sc_c0 + cb) * plane * ESS
                          : sc1b + ((size_t)n * p.sc_c1 + (cb - p.sc_c0)) * plane * ESS;
  };
  auto unit_g = [&](int i) -> int { return i < FULL ? 0 : (i < 2 * FULL ? 1 : g2); };
  // patch loads are buffer loads: descriptor = the chunk's 16 channel planes (uniform), soffset = the channel
  // plane (uniform), voffset = the lane's halo offset (fixed for the whole tile) -- no per-load address math
  int soff[8];  // byte offsets of the 8 channel planes of a k-group: loop-invariant SGPRs
#pragma unroll
  for (int j = 0; j < 8; ++j) soff[j] = __builtin_amdgcn_readfirstlane(j * plane * 4);
  // descriptor of k-group g of chunk q: its 8 channel planes / its channel block (uniform)
  auto grp_rs = [&](const char* sp, int q, int g) -> __amdgpu_buffer_rsrc_t {
    if constexpr (S2D) {  // group (cb, py, px) of the space-to-depth image: block cb, first pixel (py, px)
      const int gi = 2 * (q + qb) + g, cb = gi >> 2, pp = gi & 3;
      const int first = ((pp >> 1) * p.win + (pp & 1)) * (SB ? 8 : 1);  // in elements ([N,C,H,W]: the group's 8 planes start there)
      return __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(src0b + (((size_t)n * p.c0 + cb * 8) * plane + first) * ESS), 0, (8 * plane - first) * ESS,
          0x00020000);
    } else {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sp + (size_t)(g * 8) * plane * ESS), 0,
                                               8 * plane * ESS, 0x00020000);
    }
  };
  auto load_unit_to = [&](Patch& dst, int i, const char* sp, int q) {
    const __amdgpu_buffer_rsrc_t rs = grp_rs(sp, q, unit_g(i));
    if constexpr (S16) {  // the pixel's 8 channels are 16 contiguous bytes
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i] * 16, 0, 0);
      dst.q[i][0] = v.x; dst.q[i][1] = v.y; dst.q[i][2] = v.z; dst.q[i][3] = v.w;
    } else if constexpr (SB) {  // (the k-group's 8 planes and its channel block start at the same address)
      const float4 lo = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i] * 32, 0, 0));
      const float4 hi = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i] * 32, 16, 0));
      dst.f[i][0] = lo.x; dst.f[i][1] = lo.y; dst.f[i][2] = lo.z; dst.f[i][3] = lo.w;
      dst.f[i][4] = hi.x; dst.f[i][5] = hi.y; dst.f[i][6] = hi.z; dst.f[i][7] = hi.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        dst.f[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, goff[i] * 4, soff[j], 0));
    }
  };
  auto load_unit = [&](int i, int q, const char* sp) { load_unit_to(xr, i, sp, q); };
  // channels (2 jp, 2 jp + 1) of unit i as fp32
  auto pair_of = [&](const Patch& src, int i, int jp, float& a, float& b) {
    if constexpr (S16) {
      a = lo16<PREC>(src.q[i][jp]);
      b = hi16<PREC>(src.q[i][jp]);
    } else {
      a = src.f[i][2 * jp];
      b = src.f[i][2 * jp + 1];
    }
  };
  // the pair after GroupNorm affine (sc0, sc1, sh0, sh1) + SiLU -> operand words: (hi, scaled lo) fp16 pairs of the
  // split, or one rounded pair in the 16-bit type
  auto to_operand = [&](auto kt, float a, float b, const float4& s4, unsigned& w1, unsigned& w2) {
    if constexpr (decltype(kt)::value == 1) {  // shortcut chunk: the raw source, pre-scaled by the range guard
      if constexpr (PREC == 0) {
        a *= xs;
        b *= xs;
      }
    } else {
      if (has_ss) {
        a = a * s4.x + s4.z;
        b = b * s4.y + s4.w;
      } else if constexpr (PREC == 0 && ACT != 2) {
        a *= xs;
        b *= xs;
      }
      const float sa = silu_fast_h(a), sb = silu_fast_h(b);
      a = do_silu ? sa : a;
      b = do_silu ? sb : b;
    }
    if constexpr (PREC == 0) {
      const _Float16 a1 = (_Float16)a, b1 = (_Float16)b;
      const half2v h = {a1, b1};
      const half2v l = {(_Float16)((a - (float)a1) * 2048.0f), (_Float16)((b - (float)b1) * 2048.0f)};
      w1 = __builtin_bit_cast(unsigned, h);
      w2 = __builtin_bit_cast(unsigned, l);
    } else {
      w1 = pack2<PREC>(a, b);
      w2 = 0;
    }
  };
  typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
  auto commit_unit_from = [&](const Patch& src, int i, int q, unsigned char* buf) {  // q: chunk staged
    unsigned w1[4], w2[4];
    float4 sr[4];
    if (has_ss) {
      const float4* ssq = reinterpret_cast<const float4*>(ssl + 2 * ((q + qb) * H2_KC + unit_g(i) * 8));
#pragma unroll
      for (int j = 0; j < 4; ++j) sr[j] = ssq[j];
    }
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      if constexpr (S16 && ACT == 0) {  // 16-bit source used as it is (data-gradient convs): the word IS the operand
        w1[jp] = src.q[i][jp];
        w2[jp] = 0;
      } else {
        float a, b;
        pair_of(src, i, jp, a, b);
        to_operand(K0{}, a, b, sr[jp], w1[jp], w2[jp]);
      }
    }
    // (branch-free: a branch here would fence the instruction scheduler between staging and MFMAs)
    _Float16* xb = reinterpret_cast<_Float16*>(buf);
    *reinterpret_cast<st_u32x4*>(xb + xoff[i]) = st_u32x4{w1[0], w1[1], w1[2], w1[3]};
    if constexpr (NP == 2) *reinterpret_cast<st_u32x4*>(xb + xoff2[i]) = st_u32x4{w2[0], w2[1], w2[2], w2[3]};
  };
  // The same work in quarter-unit steps, so that a K-chunk's staging can be dealt out evenly over its taps: step P
  // turns two channels (2jp, 2jp+1) of unit P/4 into operand words -- the unit's LDS write rides on its last step --
  // and refills the registers just freed with chunk q+2's values.
  unsigned w1s[H2_NU][4], w2s[H2_NU][4];
  // s4: the step's scale/shift entry.  The caller reads it from the LDS table ONE TAP AHEAD: a table read consumed at
  // once queues behind the twelve fragment reads of the tap and stalls the wave -- MFMAs included -- for their whole
  // drain (20 such waits per K-chunk were a quarter of the loop's time).  The read is volatile so that it stays where
  // it is written: a plain load is sunk to its use, across the tap boundary, before the scheduler ever sees it.
  typedef float ss_f4 __attribute__((ext_vector_type(4)));
  auto ss_entry = [&](int P, int qs) -> float4 {
    // (the explicit LDS address space matters: a volatile access through a generic pointer is a flat load)
    const ss_f4 v = *(const volatile __attribute__((address_space(3))) ss_f4*)(
        ssl + 2 * ((qs + qb) * H2_KC + unit_g(P / 4) * 8 + 2 * (P % 4)));
    return make_float4(v.x, v.y, v.z, v.w);
  };
  constexpr int SS_NSTEP = 4 * H2_NU, SS_ST = TAPS > 1 ? TAPS - 1 : 1, SS_SPT = (SS_NSTEP + SS_ST - 1) / SS_ST;
  float4 s4b[2][KS == 3 ? SS_SPT : 1];  // [tap parity][step of the tap]
  // entries of the steps that ride on tap `tap` of the chunk that stages chunk qs
  auto load_ss = [&](int tap, int qs) {
    if constexpr (KS == 3) {
      if (has_ss) {
#pragma unroll
        for (int P = tap * SS_NSTEP / SS_ST; P < (tap + 1) * SS_NSTEP / SS_ST; ++P)
          s4b[tap & 1][P - tap * SS_NSTEP / SS_ST] = ss_entry(P, qs);
      }
    }
  };
  auto stage_step = [&](int P, int qs, unsigned char* buf, bool stage, bool load, const char* spn, const float4& s4) {  // (loads: chunk qs + 1)
    const int i = P / 4, jp = P % 4;
    if (stage) {
      if constexpr (S16 && ACT == 0) {
        w1s[i][jp] = xr.q[i][jp];
        w2s[i][jp] = 0;
      } else {
        float a, b;
        pair_of(xr, i, jp, a, b);
        to_operand(K0{}, a, b, s4, w1s[i][jp], w2s[i][jp]);
      }
      if (jp == 3) {
        _Float16* xb = reinterpret_cast<_Float16*>(buf);
        *reinterpret_cast<st_u32x4*>(xb + xoff[i]) = st_u32x4{w1s[i][0], w1s[i][1], w1s[i][2], w1s[i][3]};
        if constexpr (NP == 2)
          *reinterpret_cast<st_u32x4*>(xb + xoff2[i]) = st_u32x4{w2s[i][0], w2s[i][1], w2s[i][2], w2s[i][3]};
      }
    }
    if (load) {
      const __amdgpu_buffer_rsrc_t rs = grp_rs(spn, qs + 1, unit_g(i));
      if constexpr (S16) {  // the unit's four words are free after its last step: one 16-byte load refills them
        if (jp == 3) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i] * 16, 0, 0);
          xr.q[i][0] = v.x; xr.q[i][1] = v.y; xr.q[i][2] = v.z; xr.q[i][3] = v.w;
        }
      } else if constexpr (SB) {  // four channels are free after every second step: one 16-byte load refills them
        if (jp & 1) {
          const float4 v4 =
              __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i] * 32, 8 * (jp - 1), 0));
          xr.f[i][2 * jp - 2] = v4.x; xr.f[i][2 * jp - 1] = v4.y; xr.f[i][2 * jp] = v4.z; xr.f[i][2 * jp + 1] = v4.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e)
          xr.f[i][2 * jp + e] =
              __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, goff[i] * 4, soff[2 * jp + e], 0));
      }
    }
  };
  auto commit_unit = [&](int i, int q, unsigned char* buf) { commit_unit_from(xr, i, q, buf); };
  // weight slab of chunk q: 36 segments (piece, tap, g) of 64 couts x 16 B, moved global -> LDS by DMA;
  // wave w moves segments w, w+4, ...
  // Addressing is split so that a DMA costs one 64-bit scalar add: the tile's weight base and the byte offset of each
  // of this wave's segments are loop-invariant scalars, the chunk offset is added once per chunk by the caller, and
  // the only vector operand is the constant lane * 16.
  const unsigned segb = (unsigned)p.wh_stride * 16u;  // bytes of one (piece, tap, g) segment row in global memory
  const unsigned chunkb = G::NSEG * segb;             // bytes of one K-chunk's weights
  const char* wtile = static_cast<const char*>(p.wh) +
                      (((size_t)phase * (p.cin / H2_KC) + qb) * G::NSEG * p.wh_stride + m0) * 16;
  // a DMA moves 1 KB = 64 / BM segments of BM couts x 16 B: LDS [segment][cout][8 halfs] is contiguous, in global
  // memory the segments are `segb` apart
  // (BM = 32: a unit is two segments of 32 couts; BM = 128: a segment is two units of 64 couts)
  constexpr int SPU = BM >= 64 ? 1 : 64 / BM, UPS = BM >= 64 ? BM / 64 : 1;
  int segoff[G::NDMA];
#pragma unroll
  for (int k = 0; k < G::NDMA; ++k) {
    const int u = min(wave + NW * k, G::NUNIT - 1);
    segoff[k] = __builtin_amdgcn_readfirstlane((u / UPS) * SPU * (int)segb + (u % UPS) * 1024);
  }
  const int lane16 = (lane % (BM < 64 ? BM : 64)) * 16 + (BM < 64 ? (lane / BM) * (int)segb : 0);
  auto dma_weights = [&](int k, const char* wq, unsigned char* buf) {  // wq: wtile + chunk * chunkb (uniform)
    // (uniform; a wave whose last share falls past the end repeats the final unit: same bytes, no branch)
    const int unit = min(wave + NW * k, G::NUNIT - 1);
    // Issued as inline asm on purpose: hipcc's wait-count pass cannot tell the DMA's LDS destination (the other
    // buffer) from the fragment reads of this one, and with a DMA it knows of in flight it puts vmcnt(0) -- a wait
    // for every outstanding patch load as well -- in front of each following ds_read.  Untracked VMEM operations
    // only make the compiler's own counted vmcnt(N) waits stricter (the counter retires in order); the DMA's
    // completion is waited for explicitly before the chunk's closing barrier.
    const unsigned lds_addr =
        (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(buf + unit * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n" ::"v"(lane16),
                 "s"(PRE ? sgpr_ptr(wq + segoff[k]) : wq + segoff[k]),
                 "s"(__builtin_amdgcn_readfirstlane(lds_addr))  // (uniform by construction)
                 : "memory");
  };

  // the shortcut's weight slab of a chunk: (piece, g) segments of BM couts x 16 B = NUNIT_SC 1-KB units at the start of the
  // W region, [piece][tap 1][g][cout][8]
  constexpr int NUNIT_SC = 2 * NP * BM / 64, NDMA_SC = (NUNIT_SC + NW - 1) / NW;
  const unsigned sc_segb = (unsigned)p.sc_wh_stride * 16u, sc_chunkb = 2 * NP * sc_segb;
  const char* sc_wtile = SC ? static_cast<const char*>(p.sc_wh) + (size_t)m0 * 16 : nullptr;
  int segoff_sc[NDMA_SC];
#pragma unroll
  for (int k = 0; k < NDMA_SC; ++k) {
    const int u = min(wave + NW * k, NUNIT_SC - 1);
    segoff_sc[k] = __builtin_amdgcn_readfirstlane((u / UPS) * SPU * (int)sc_segb + (u % UPS) * 1024);
  }
  const int lane16_sc = (lane % (BM < 64 ? BM : 64)) * 16 + (BM < 64 ? (lane / BM) * (int)sc_segb : 0);
  auto dma_weights_sc = [&](int k, const char* wq, unsigned char* buf) {  // wq: sc_wtile + chunk * sc_chunkb (uniform)
    const int unit = min(wave + NW * k, NUNIT_SC - 1);
    const unsigned lds_addr =
        (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(buf + unit * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n" ::"v"(lane16_sc), "s"(wq + segoff_sc[k]),
                 "s"(__builtin_amdgcn_readfirstlane(lds_addr))
                 : "memory");
  };

  // PRE: the chunk's patch as NPU 1-KB DMA units; wave w moves units w, w + NW, ...  Unit U = (region r = U / PUNITS,
  // u = U % PUNITS): LDS [r][64 u ..] <- padded image, piece r / 2, channel block 2 q + (r & 1), halo positions 64 u + lane
  constexpr int NPU = NP * 2 * G::PUNITS, NPD = PRE ? (NPU + NW - 1) / NW : 1;
  int pvoff[NPD], pldso[NPD];
  size_t pgoff[NPD];
  const int pre_wp = p.win + 2;                                  // padded row length (the fold mode tiles the source grid too)
  const size_t pre_blk = (size_t)(p.hin + 2) * pre_wp * 16;      // bytes of one padded channel block
  const char* pre_tile = nullptr;                                // chunk 0, piece 0, k-group 0, first halo position of this tile
  if constexpr (PRE) {
#pragma unroll
    for (int k = 0; k < NPD; ++k) {
      const int U = min(wave + NW * k, NPU - 1), r = U / G::PUNITS, u = U - r * G::PUNITS;
      const int pos = min(u * 64 + lane, H2_PSZ - 1);            // (the last unit's spare lanes re-fetch the last position)
      const int py = pos / H2_PW, px = pos - py * H2_PW;
      pvoff[k] = (py * pre_wp + px) * 16;
      pldso[k] = __builtin_amdgcn_readfirstlane((r * H2_PSTR + u * 64) * 16);
      const size_t go = (size_t)(r >> 1) * p.pre_piece_stride + (size_t)(r & 1) * pre_blk;
      pgoff[k] = ((size_t)__builtin_amdgcn_readfirstlane((unsigned)(go >> 32)) << 32) |
                 (size_t)__builtin_amdgcn_readfirstlane((unsigned)go);
    }
    // (padded coordinates: halo position (py, px) of the tile at (oy0, ox0) is padded pixel (oy0 + py, ox0 + px))
    pre_tile = static_cast<const char*>(p.pre) +
               ((size_t)n * (p.cin / 8) + 2 * (size_t)qb) * pre_blk + ((size_t)oy0 * pre_wp + ox0) * 16;
  }
  auto dma_patch = [&](int k, const char* pq, unsigned char* buf) {  // pq: pre_tile + chunk * 2 * pre_blk (uniform)
    const unsigned lds_addr =
        (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(buf + H2_WHALFS * 2 + pldso[k]);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n" ::"v"(pvoff[k]), "s"(sgpr_ptr(pq + pgoff[k])),
                 "s"(__builtin_amdgcn_readfirstlane(lds_addr))
                 : "memory");
  };

  f32x16 acc_hi[MTN][NT], acc_lo[NP == 2 ? MTN : 1][NP == 2 ? NT : 1];  // (acc_lo: the split's scaled low-order products)
#pragma unroll
  for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc_hi[mt][nt][r] = 0.f;
        if constexpr (NP == 2) acc_lo[mt][nt][r] = 0.f;
      }

  unsigned char* buf0 = smem_raw;
  unsigned char* buf1 = smem_raw + H2_BUF1_OFF;

  auto scale_acc = [&](float f) {
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc_hi[mt][nt][r] *= f;
          if constexpr (NP == 2) acc_lo[mt][nt][r] *= f;
        }
  };
  // ---- Fused shortcut: sc_cin / 16 one-tap chunks of the resnet's raw input on the same accumulators -----------------
  // No staging pass here.  A 1x1 conv's B fragment -- the 8 channels of k-group `half` at this lane's pixel -- is used by
  // exactly one wave, so the fp32 -> (hi, scaled lo) split is done by the consumer, on fragments read straight from the RAW
  // fp32 rows: in the channel-blocked layout a tile row of one channel block is 1 KB of contiguous HBM, i.e. ONE
  // global_load_lds_dwordx4 per (k-group, row), issued by the wave that will read it.  The K loop's buffers are free, so
  // the rows go into a ring of SCD chunk slots [g][row][32 px][8 ch] fp32 (+ the shortcut's weight slabs beside it):
  // SCD - 1 chunks of loads are in flight per workgroup with no register holding them -- the phase streams at what HBM
  // gives (a register-staged one-tap chunk has one chunk's time, ~0.4 us, to cover ~1 us of latency: measured 2.3-3.0 k
  // cycles per chunk).  One barrier per chunk publishes the weight slab; raw rows need none (own DMAs, own vmcnt).
  // Which phase comes first alternates with the tile's position (16-row band + column tile: the same for every tile
  // geometry and batch size, so a pixel's summation order never depends on the launch): while one workgroup streams its
  // shortcut rows -- HBM-bound, matrix pipe idle -- its neighbours are in their 3x3 chunks -- matrix pipe busy, HBM idle.
  // With every workgroup in the same phase at the same time the two phases' times simply add up (measured: 252 us of
  // 3x3 chunks + 149 us of shortcut rows at 5.4 TB/s on the 192 -> 64 @ 256^2 resnets).
  const bool sc_first = SC && (((oy0 >> 4) + tx) & 1) != 0;
  auto sc_phase = [&](const bool first) {
    constexpr int SCD = 4;                               // ring depth
    // 16-bit tensors: a row of a channel block is 512 B, so ONE DMA brings both k-groups of a row (lanes 32-63 address
    // the second block: slot layout [row][g][32 px][8 ch]) and the words it lands ARE the B operands -- no arithmetic at all
    constexpr int NRAW = S16 ? NT : 2 * NT;              // raw-row DMAs per wave and chunk
    constexpr int RAW_BYTES = NW * NRAW * 1024;          // a chunk's raw rows
    constexpr int SCW_BYTES = NUNIT_SC * 1024;           // a chunk's weight slab
    constexpr int SCW_OFF = SCD * RAW_BYTES;
    constexpr int PER = NDMA_SC + NRAW;                  // DMA instructions per wave and chunk
    static_assert(SCD == 4, "the tail below is written out for three chunks in flight");
    static_assert(!SC || SCW_OFF + SCD * SCW_BYTES <= 160 * 1024 / OCC, "the shortcut ring must fit the workgroup's share of LDS");
    static_assert(!SC || (SCD - 2) * PER <= 63, "vmcnt range");
    // this workgroup's shortcut chunks [sb, sb + ns): all of them, or -- split-K (gridDim.y slices of a small grid) -- an
    // equal share, like its 3x3 chunks (the host dispatches here with at least SCD chunks per slice)
    int ns = p.sc_cin / H2_KC, sb = 0;
    if (gridDim.y > 1) {
      const int per = (ns + (int)gridDim.y - 1) / (int)gridDim.y;
      sb = (int)blockIdx.y * per;
      ns = min(per, ns - sb);
    }
    int rowoff[NRAW], ldsoff[NRAW];                      // this wave's pieces: global / LDS byte offsets (uniform)
#pragma unroll
    for (int k = 0; k < NRAW; ++k) {
      const int g = S16 ? 0 : k / NT, row = wave * NT + k % NT;
      rowoff[k] = __builtin_amdgcn_readfirstlane(g * 8 * plane * ESS + ((oy0 + row) * p.win + ox0) * 8 * ESS);
      ldsoff[k] = __builtin_amdgcn_readfirstlane(S16 ? row * 1024 : (g * H2_TH + row) * 1024);
    }
    // per-lane part of a raw-row address: 16 bytes per lane; 16-bit rows: lanes 32-63 read the chunk's second channel block
    const int lane_raw = S16 ? (lane >> 5) * (8 * plane * ESS) + (lane & 31) * 16 : lane * 16;
    auto sc_issue = [&](int j) {  // every DMA of chunk j (uniform)
      unsigned char* slab = smem_raw + SCW_OFF + (j & (SCD - 1)) * SCW_BYTES;
#pragma unroll
      for (int k = 0; k < NDMA_SC; ++k) dma_weights_sc(k, sc_wtile + (size_t)(sb + j) * sc_chunkb, slab);
      const char* sp = sc_src_of(sb + j);
      unsigned char* ring = smem_raw + (j & (SCD - 1)) * RAW_BYTES;
#pragma unroll
      for (int k = 0; k < NRAW; ++k) {
        const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(ring + ldsoff[k]);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n" ::"v"(lane_raw), "s"(sp + rowoff[k]),
                     "s"(__builtin_amdgcn_readfirstlane(lds_addr))
                     : "memory");
      }
    };
    auto sc_compute = [&](int j) {
      const _Float16* wl = reinterpret_cast<const _Float16*>(smem_raw + SCW_OFF + (j & (SCD - 1)) * SCW_BYTES);
      const unsigned char* rawb = smem_raw + (j & (SCD - 1)) * RAW_BYTES;
      const float* raw = reinterpret_cast<const float*>(rawb);
      half8 fa[MTN][NP];
      float4 rv[S16 ? 1 : NT][2];
      half8 rh[S16 ? NT : 1];
#pragma unroll
      for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
          fa[mt][pc] = *reinterpret_cast<const half8*>(wl + ((pc * 2 + half) * BM + mt * 32 + l31) * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr (S16) {
          rh[nt] = *reinterpret_cast<const half8*>(rawb + ((wave * NT + nt) * 2 + half) * 512 + l31 * 16);
        } else {
          const float4* rp = reinterpret_cast<const float4*>(raw + ((half * H2_TH + wave * NT + nt) * 32 + l31) * 8);
          rv[nt][0] = rp[0];
          rv[nt][1] = rp[1];
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        half8 b_hi, b_lo;
        if constexpr (S16) {
          b_hi = rh[nt];
          b_lo = rh[nt];
        } else {
          unsigned w1[4], w2[4];
          const float4 dummy = make_float4(1.f, 1.f, 0.f, 0.f);
          to_operand(K1{}, rv[nt][0].x, rv[nt][0].y, dummy, w1[0], w2[0]);
          to_operand(K1{}, rv[nt][0].z, rv[nt][0].w, dummy, w1[1], w2[1]);
          to_operand(K1{}, rv[nt][1].x, rv[nt][1].y, dummy, w1[2], w2[2]);
          to_operand(K1{}, rv[nt][1].z, rv[nt][1].w, dummy, w1[3], w2[3]);
          b_hi = __builtin_bit_cast(half8, st_u32x4{w1[0], w1[1], w1[2], w1[3]});
          b_lo = __builtin_bit_cast(half8, st_u32x4{w2[0], w2[1], w2[2], w2[3]});
        }
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt) {
          if constexpr (NP == 2) {
            acc_hi[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][0], b_hi, acc_hi[mt][nt], 0, 0, 0);
            acc_lo[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][0], b_lo, acc_lo[mt][nt], 0, 0, 0);
            acc_lo[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][1], b_hi, acc_lo[mt][nt], 0, 0, 0);
          } else {
            acc_hi[mt][nt] = mma16<PREC>(fa[mt][0], b_hi, acc_hi[mt][nt]);
          }
        }
      }
    };
    // (first phase: nothing has touched LDS yet; second: the K loop's closing barrier is behind us)
#pragma unroll
    for (int j = 0; j < SCD - 1; ++j) sc_issue(j);
    if constexpr (PREC == 0) {
      // Range guard: the shortcut's products arrive scaled by xs = 2^-e (their source was).  Second phase: the 3x3
      // chunks' sums, already in the accumulators, take the same factor here and the epilogue multiplies everything by
      // 2^e.  First phase: the factor is taken back out after the last shortcut chunk (below), before the 3x3 chunks add
      // theirs.  Powers of two, exact; uniform branches that only an out-of-range source ever takes.
      if (!first && xs != 1.f) scale_acc(xs);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SCD - 2) * PER) : "memory");  // chunk 0 has landed
    __syncthreads();
    int j = 0;
    for (; j + SCD - 1 < ns; ++j) {
      sc_issue(j + SCD - 1);   // into the slot chunk j - 1 was read from (own rows: program order; slab: the barrier)
      sc_compute(j);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SCD - 2) * PER) : "memory");  // chunk j + 1 has landed
      __syncthreads();
    }
    // the last SCD - 1 chunks: nothing left to issue
    sc_compute(j++);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    __syncthreads();
    sc_compute(j++);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    sc_compute(j);
    if constexpr (PREC == 0) {
      if (first && xs != 1.f) scale_acc(rg_out);
    }
    __syncthreads();  // (the K loop's prologue / the epilogue re-use LDS)
  };
  if constexpr (SC) {
    if (sc_first) sc_phase(true);
  }

  // zero padding: halo positions outside the image are zeroed once in both buffers and never written again
  // (PRE: the operand image carries its own zero border)
#pragma unroll
  for (int i = 0; i < (PRE ? 0 : H2_NU); ++i) {
    if (zoff[i] >= 0) {
      half8 z;
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
      _Float16* b0 = reinterpret_cast<_Float16*>(buf0);
      _Float16* b1 = reinterpret_cast<_Float16*>(buf1);
      *reinterpret_cast<half8*>(b0 + zoff[i]) = z;
      *reinterpret_cast<half8*>(b1 + zoff[i]) = z;
      if constexpr (NP == 2) {
        *reinterpret_cast<half8*>(b0 + zoff[i] + 2 * H2_PSTR * 8) = z;
        *reinterpret_cast<half8*>(b1 + zoff[i] + 2 * H2_PSTR * 8) = z;
      }
    }
  }
  // prologue: chunk 0 -> buffer 0; chunk 1 -> registers.  Everything that goes to memory is issued first and
  // together (both chunks' patches, the weight DMAs, the scale/shift table), so the tile pays one memory round
  // trip before its first MFMA, not one per dependent step.
#ifdef DSG_H2_TIMING
  unsigned long long rt_p[4];
#define DSG_PT(i) do { __builtin_amdgcn_sched_barrier(0); rt_p[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
  DSG_PT(0);
#else
#define DSG_PT(i)
#endif
  if constexpr (PRE) {  // chunk 0: weights and patch by DMA, nothing else
#pragma unroll
    for (int k = 0; k < G::NDMA; ++k) dma_weights(k, wtile, buf0);
#pragma unroll
    for (int k = 0; k < NPD; ++k) dma_patch(k, pre_tile, buf0);
    DSG_PT(1);
    DSG_PT(2);
    DSG_PT(3);
  } else {
    // Issue order: weight DMAs, chunk 0's patch (into a scratch set), chunk 1's patch (into xr, where the K loop
    // expects it).  The tile waits only for the DMAs and chunk 0 -- vmcnt retires in order, so a counted wait with
    // chunk 1's loads still outstanding covers exactly those -- and chunk 1 lands under chunk 0's MFMAs.
    Patch xr0;
    float ssv[2048 / NTH];  // this thread's share of the image's scale/shift table (cin <= 1024): oldest loads
    if (has_ss) {
#pragma unroll
      for (int k = 0; k < 2048 / NTH; ++k) ssv[k] = ssg[min(tid + NTH * k, 2 * p.cin - 1)];
    }
#pragma unroll
    for (int k = 0; k < G::NDMA; ++k) dma_weights(k, wtile, buf0);
    const char* sp = src_of(0);
#pragma unroll
    for (int i = 0; i < H2_NU; ++i) load_unit_to(xr0, i, sp, 0);
    if (nq > 1) {
      const char* sp1 = src_of(1);
#pragma unroll
      for (int i = 0; i < H2_NU; ++i) load_unit_to(xr, i, sp1, 1);
    }
    DSG_PT(1);
    if (has_ss) {
#pragma unroll
      for (int k = 0; k < 2048 / NTH; ++k)
        if (tid + NTH * k < 2 * p.cin) {
          // global [c][scale | shift] -> LDS per channel PAIR (sc0, sc1, sh0, sh1): a staging step's two channels
          // then take their scales and shifts as register pairs (one packed FMA, no shuffling moves)
          const int idx = tid + NTH * k, c = idx >> 1, which = idx & 1;
          ssl[4 * (c >> 1) + 2 * which + (c & 1)] = ssv[k];
        }
      __syncthreads();  // the scale/shift table is in LDS
    }
    DSG_PT(2);
#pragma unroll
    for (int i = 0; i < H2_NU; ++i) commit_unit_from(xr0, i, 0, buf0);
    DSG_PT(3);
  }
  // (the DMAs were issued before every patch load: once chunk 0's values have been used they have landed; 8 * NU
  // loads of chunk 1 may still be in flight)
  if (nq > 1 && !PRE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S16 ? 1 : (SB ? 2 : 8)) * H2_NU) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // One K-chunk: MFMAs on `cur`; STAGE: chunk q+1 (patch in registers, weights by DMA) goes into `nxt`;
  // LOAD: chunk q+2's patch is fetched into the registers just freed.
#ifdef DSG_H2_TIMING
  unsigned long long t_tap[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_vm = 0, t_bar = 0;
  const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();  // 100 MHz, same base on every CU
#endif
  auto chunk = [&](int q, auto stage_tag, auto load_tag) {
    constexpr bool STAGE = decltype(stage_tag)::value, LOAD = decltype(load_tag)::value;
    unsigned char* cur = (q & 1) ? buf1 : buf0;
    unsigned char* nxt = (q & 1) ? buf0 : buf1;
    const char* spn = (LOAD && !PRE) ? src_of(q + 2) : nullptr;
    const char* wqn = wtile + (size_t)(q + 1) * chunkb;  // the staged chunk's weights
    const char* pqn = PRE ? pre_tile + (size_t)(q + 1) * 2 * pre_blk : nullptr;  // ... and its pre-staged patch
    const _Float16* wl = reinterpret_cast<const _Float16*>(WS ? buf0 : cur);
    const _Float16* xl = reinterpret_cast<const _Float16*>(cur) + H2_WHALFS;
    // Operand fragments are fetched one tap ahead into the other half of fa/fb: the reads of tap t+1 are issued
    // BEFORE tap t's staging writes in program order, so tap t's MFMAs depend on registers only and the scheduler
    // is free to interleave them with the staging work (LDS reads after a possibly-aliasing LDS write are not).
    half8 fa[2][MTN][NP], fb[2][NT][NP];  // [parity][tile][piece]
    auto load_frags = [&](int tap, int par) {
      // folded up-sampler: the phase's 2x2 corner of the patch; stride 2: the {y-1, y} x {x-1, x} corner
      // GM 4 (data gradient of nearest-2x + 3x3 conv: a 4x4 stride-2 window over dY that starts at (2y - 1, 2x - 1)): window
      // row i is source row 2 (y - 1 + (i + 1) / 2) + ((i + 1) & 1), so the chunk's row parity py = (q + qb) & 1 owns window
      // rows {1 - py, 3 - py} = patch rows (tap >> 1) + 1 - py (the patch starts at y - 1); same in x with px = the lane half
      const int dy = GM == 2 ? (phase >> 1) + (tap >> 1) : (GM == 3 ? (tap >> 1) : (GM == 4 ? (tap >> 1) + 1 - ((q + qb) & 1) : tap / KS));
      const int dx = GM == 2 ? (phase & 1) + (tap & 1) : (GM == 3 ? (tap & 1) : (GM == 4 ? (tap & 1) + 1 - half : tap % KS));
#ifdef DSG_H2_ABL_NOFA   // (tools/ timing experiment: tap 0's weight fragments serve every tap -- wrong values, the loop without 8/9 of the A reads)
      if (tap == 0) {
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
          for (int pc = 0; pc < NP; ++pc)
            fa[0][mt][pc] = fa[1][mt][pc] =
                *reinterpret_cast<const half8*>(wl + (((pc * TAPS + tap) * 2 + half) * BM + mt * 32 + l31) * 8);
      }
#else
#pragma unroll
      for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
          fa[par][mt][pc] =
              *reinterpret_cast<const half8*>(wl + (((pc * TAPS + tap) * 2 + half) * BM + mt * 32 + l31) * 8);
#endif
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
          fb[par][nt][pc] = *reinterpret_cast<const half8*>(
              xl + ((pc * 2 + half) * H2_PSTR + (wave * NT + nt + dy) * H2_PW + l31 + dx) * 8);
    };
    load_frags(0, 0);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      __builtin_amdgcn_sched_barrier(0);
#ifdef DSG_H2_TIMING
      const unsigned long long tt0 = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
#endif
      // next tap's scale/shift entries (BEFORE the fragments in program order: LDS returns in order, so a counted wait
      // covers the entries alone); the clear last tap fetches tap 0's entries of the next chunk
      if (!PRE && STAGE && tap + 1 < TAPS - 1) load_ss(tap + 1, q + 1);
      if (!PRE && LOAD && tap == TAPS - 1) load_ss(0, q + 2);
      if (tap + 1 < TAPS) load_frags(tap + 1, (tap + 1) & 1);
      if constexpr (PRE) {
        // the staged chunk's DMAs -- patch units first, then the weight units -- dealt out over taps 0..TAPS-2 (all of the
        // patch units on the first tap measured slower: -0.6 % against +0.7 % on the step)
        if (STAGE && tap < TAPS - 1) {
          constexpr int ND = NPD + G::NDMA, ST = TAPS - 1;
#pragma unroll
          for (int d = tap * ND / ST; d < (tap + 1) * ND / ST; ++d) {
            if (d < NPD) dma_patch(d, pqn, nxt);
            else dma_weights(d - NPD, wqn, nxt);
          }
        }
      }
      if (KS == 1) {  // one tap: all units and the four weight segments ride on it
#pragma unroll
        for (int u = 0; u < H2_NU; ++u) {
          if (STAGE) commit_unit(u, q + 1, nxt);
          if (LOAD) load_unit(u, q + 2, spn);
        }
        if (STAGE) dma_weights(0, wqn, nxt);  // (KS = 1: NSEG = 4 <= NW)
      }
      // KS = 3: the chunk's staging steps and weight DMAs are dealt out evenly over taps 0..TAPS-2 (the last tap
      // stays clear so that the newest loads have a tap's worth of MFMAs to land before the closing vmcnt(0))
      if (KS == 3 && !PRE && tap < TAPS - 1) {
        constexpr int NSTEP = 4 * H2_NU, ST = TAPS > 1 ? TAPS - 1 : 1;
#pragma unroll
        for (int P = tap * NSTEP / ST; P < (tap + 1) * NSTEP / ST; ++P)
          stage_step(P, q + 1, nxt, STAGE, LOAD, spn,
                     has_ss ? s4b[tap & 1][P - tap * NSTEP / ST] : make_float4(1.f, 1.f, 0.f, 0.f));
#ifndef DSG_H2_ABL_NODMA
        if (STAGE && !WS) {
#pragma unroll
          for (int k = tap * G::NDMA / ST; k < (tap + 1) * G::NDMA / ST; ++k) dma_weights(k, wqn, nxt);
        }
#endif
      }
      const int par = tap & 1;
#pragma unroll
      for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (NP == 2) {
            acc_hi[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][mt][0], fb[par][nt][0], acc_hi[mt][nt], 0, 0, 0);
            acc_lo[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][mt][0], fb[par][nt][1], acc_lo[mt][nt], 0, 0, 0);
            acc_lo[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][mt][1], fb[par][nt][0], acc_lo[mt][nt], 0, 0, 0);
          } else {
            acc_hi[mt][nt] = mma16<PREC>(fa[par][mt][0], fb[par][nt][0], acc_hi[mt][nt]);
          }
        }
      // Issue order within the tap: with one wave per SIMD nothing else fills the matrix pipe while this wave
      // issues staging work, so spread that work between the MFMAs (at most ~5 issues hide behind one MFMA)
      // instead of leaving it in one block as the scheduler would.
      if (KS == 3 && !PRE && tap < TAPS - 1 && (STAGE || LOAD)) {
#pragma unroll
        for (int m = 0; m < (NP == 2 ? 3 : 1) * MTN * NT; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
          // 2 VALU per MFMA (a third of the MFMAs: 5; 128 couts, twice the MFMAs per tap again: 3)
          __builtin_amdgcn_sched_group_barrier(0x002, NP == 2 ? 2 : (BM == 128 ? 3 : 5), 0);
#ifdef DSG_H2_SPREAD_READS
          // ... and the next tap's fragment reads dealt out over this tap's MFMAs (left alone the scheduler bunches them
          // at the tap's end: four waves' 48 KB leave together and the next tap's first MFMAs wait for them in turn)
          {
            constexpr int NM = (NP == 2 ? 3 : 1) * MTN * NT, NR = (MTN + NT) * NP;
            if ((m + 1) * NR / NM > m * NR / NM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if ((m + 1) * NR / NM > m * NR / NM + 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#endif
        }
      }
#ifdef DSG_H2_TIMING
      __builtin_amdgcn_sched_barrier(0);
      t_tap[tap] += __builtin_readcyclecounter() - tt0;
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    auto dma_next_slab = [&]() {  // WS: the staged chunk's whole weight slab, after everyone is done with this one's
#pragma unroll
      for (int k = 0; k < G::NDMA; ++k) dma_weights(k, wqn, buf0);
    };
#ifdef DSG_H2_TIMING  // tools/ only: where does a wave wait at the end of a chunk?  (p.stats = 4 counters)
    const unsigned long long ta = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long tb = __builtin_readcyclecounter();
    __builtin_amdgcn_s_barrier();
    const unsigned long long tc = __builtin_readcyclecounter();
    t_vm += tb - ta;
    t_bar += tc - tb;
    if constexpr (WS) {  // (timing build: the exposed slab fetch counts as memory wait)
      if (STAGE) {
        dma_next_slab();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        t_vm += __builtin_readcyclecounter() - tc;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#else
    if constexpr (WS) {
      __syncthreads();  // everyone is done with the weight slab (and with cur); nxt's patch is complete
#ifndef DSG_H2_ABL_WS_NOSLAB   // (tools/ timing experiment: chunk 0's weights serve every chunk -- wrong values, the loop without the exposed slab fetch)
      if (STAGE) {
        dma_next_slab();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // the next chunk's weights are in
      }
#endif
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the weight DMAs (not tracked by the compiler) have landed
      __syncthreads();  // nxt is complete; everyone is done reading cur
    }
#endif
  };
  //
  if (nq > 1 && !PRE) load_ss(0, 1);  // tap 0 of the first chunk stages chunk 1
#ifdef DSG_H2_TIMING
  const unsigned long long t_begin = __builtin_readcyclecounter();
  const unsigned long long rt_loop = __builtin_amdgcn_s_memrealtime();
#endif
  using T = std::true_type;
  using F = std::false_type;
  int q = 0;
#if defined(DSG_H2_ABL_NOSTAGE)  // (tools/ timing experiments only: wrong results, loop time without a component)
  for (; q + 2 < nq; ++q) chunk(q, F{}, T{});
#elif defined(DSG_H2_ABL_NOLOAD)
  for (; q + 2 < nq; ++q) chunk(q, T{}, F{});
#elif defined(DSG_H2_ABL_MFMAONLY)
  for (; q + 2 < nq; ++q) chunk(q, F{}, F{});
#else
  for (; q + 2 < nq; ++q) chunk(q, T{}, T{});
#endif
  if (q + 1 < nq) chunk(q++, T{}, F{});  // last staged chunk: nothing left to load
  chunk(q, F{}, F{});                    // last chunk: MFMAs only

  if constexpr (SC) {
    if (!sc_first) sc_phase(false);
  }
#ifdef DSG_H2_TIMING
  const unsigned long long rt_loop_end = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t_loop_cycles = __builtin_readcyclecounter() - t_begin;
#endif

  // Epilogue.  All global accesses are buffer operations on descriptors that start at this tile's first output
  // channel: the per-lane offset (row, column, +4 channels for the upper half-wave) is one VGPR computed once, the
  // (channel, row) part of each access is a scalar offset, and channels past cout fall outside the descriptor's
  // range -- loads return 0, stores are dropped -- so there is neither address arithmetic nor a bounds branch per
  // element.  The residual values of a 32-channel slab are all in flight before the first use (with one wave per
  // SIMD a load->add->store chain per element would expose the memory latency 64 times).
  // (the host only dispatches here when cout % 8 == 0, so a 4-row half-group is never split by cout)
  const bool has_r = GNB ? true : p.res != nullptr;   // (GNB: the "residual" slot carries the pre-norm x, read but not added)
  const int oscale = GM == 2 ? 2 : 1;  // folded mode: the output map is twice the tiled (low-resolution) grid
  const int oplane = p.hout * p.wout * oscale * oscale;
#ifdef DSG_H2_TIMING_NOSTATS
  const bool want_stats = false;  // (timing experiment: the record buffer is p.stats, the statistics path stays off)
#else
  const bool want_stats = p.stats != nullptr;
#endif
  float* red = reinterpret_cast<float*>(smem_raw);  // [wave][sum | sumsq][cout 64] (the K loop is done with LDS)
  constexpr int RED_FLOATS = NW * (NT / 2) * 2 * BM;
  // Cross-lane sums of the statistics go through LDS, not DPP chains: a lane leaves its row-pair values in a
  // wave-private [value][lane] table, then lane L adds up the 32 entries of table row L (and L + 64) in a fixed order.
  // (The DPP version -- five dependent adds per value, 128 values per lane and slab -- was 3.3 of the epilogue's 8.2 us.)
  constexpr int SV = 16 * (NT / 2) * 2;  // values per lane and slab: 16 channels x row pairs x (sum, sum of squares)
  // (table row r = (value, half-wave) = 32 floats; its 16-byte chunks are stored at chunk ^ (r & 7), so that both the
  // 4-byte writes of a half-wave and the 16-byte reads of eight neighbouring rows spread over all banks)
  float* stab = red + RED_FLOATS + wave * (SV * 64);
  const int stab_w = half * 32 + (l31 & 3);  // this lane's fixed part of a write address
  // row r = 2 v + half, so (r & 7) = 2 (v & 3) | half: the swizzled chunk is ((l31 >> 2) ^ half) ^ 2 (v & 3) -- four
  // per-lane values, not one per table value
  int stab_sw[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) stab_sw[c] = stab_w + (((((l31 >> 2) ^ half) ^ (2 * c)) & 7) << 2);
  const int nvalid = min(BM, p.cout - m0);        // output channels of this tile that exist
  const size_t tile_off = ((size_t)n * p.cout + m0) * oplane * ESD;  // bytes
  const int range = nvalid * oplane * ESD;
  char* dstb = static_cast<char*>(p.dst) + (size_t)blockIdx.y * p.split_stride;
  const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(dstb + tile_off, 0, range, 0x00020000);
  const char* res_base = static_cast<const char*>(p.res);
  size_t res_off = tile_off;
  int gnb_seam = BM;  // GNB: the tile's first channel that lives in gnb_x1 (relative to m0; a multiple of 32: the host's rule)
  if constexpr (GNB) {  // this channel tile's rows of x: in gnb_x0 (gnb_c0 channels per image) or gnb_x1 (the rest)
    const bool in0 = m0 < p.gnb_c0;
    res_base = static_cast<const char*>(in0 ? p.gnb_x0 : p.gnb_x1);
    res_off = ((size_t)n * (in0 ? p.gnb_c0 : p.cout - p.gnb_c0) + (in0 ? m0 : m0 - p.gnb_c0)) * oplane * ESD;
    if (in0 && p.gnb_x1 != nullptr && p.gnb_c0 < m0 + BM) gnb_seam = p.gnb_c0 - m0;  // the two tensors meet inside this tile
  }
  const __amdgpu_buffer_rsrc_t res_rs0 = __builtin_amdgcn_make_buffer_rsrc(
      has_r ? const_cast<char*>(res_base) + res_off : dstb, 0, has_r ? range : 0, 0x00020000);
  // ... and the 32-channel slabs from the seam on: gnb_x1, based so that the tile-relative channel offsets below still apply
  // (the base lies gnb_seam planes in front of the image's first x1 channel; nothing below the seam is read through it)
  const __amdgpu_buffer_rsrc_t res_rs1 =
      (GNB && gnb_seam < BM)
          ? __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p.gnb_x1)) +
                                                  ((ptrdiff_t)n * (p.cout - p.gnb_c0) - gnb_seam) * (ptrdiff_t)oplane * ESD,
                                              0, range, 0x00020000)
          : res_rs0;
  const __amdgpu_buffer_rsrc_t gss_rs = __builtin_amdgcn_make_buffer_rsrc(
      GNB ? const_cast<float*>(p.gnb_ss + ((size_t)n * p.cout + m0) * 2) : reinterpret_cast<float*>(dstb), 0,
      GNB ? nvalid * 8 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t bias_rs = __builtin_amdgcn_make_buffer_rsrc(
      p.bias ? const_cast<float*>(p.bias + m0) : reinterpret_cast<float*>(dstb), 0, p.bias ? nvalid * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t temb_rs = __builtin_amdgcn_make_buffer_rsrc(
      p.temb ? const_cast<float*>(p.temb + (size_t)n * p.temb_stride + m0) : reinterpret_cast<float*>(dstb), 0,
      p.temb ? nvalid * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t scb_rs = __builtin_amdgcn_make_buffer_rsrc(
      (SC && p.sc_bias) ? const_cast<float*>(p.sc_bias + m0) : reinterpret_cast<float*>(dstb), 0,
      (SC && p.sc_bias) ? nvalid * 4 : 0, 0x00020000);
  int voff[NT];  // bytes, per lane and row; the channel part of an address is the scalar offset
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if constexpr (DB) {  // [C/8][H][W][8]: pixel * 32 bytes + this half-wave's four channels
      const int oy = GM == 2 ? 2 * (oy0 + wave * NT + nt) + (phase >> 1) : oy0 + wave * NT + nt;
      const int ox = GM == 2 ? 2 * (ox0 + l31) + (phase & 1) : ox0 + l31;
      voff[nt] = (ox0 + l31 < p.wout) ? ((oy * (p.wout * oscale) + ox) * 8 + 4 * half) * ESD : 0x7FFFFFF0;
    } else {
      voff[nt] = GM == 2 ? (ox0 + l31 < p.wout ? (4 * half * oplane + (2 * (oy0 + wave * NT + nt) + (phase >> 1)) * (2 * p.wout) +
                                                   2 * (ox0 + l31) + (phase & 1)) * 4 : 0x7FFFFFF0)
                         : (ox0 + l31 < p.wout ? (4 * half * oplane + (oy0 + wave * NT + nt) * p.wout + ox0 + l31) * 4
                                               : 0x7FFFFFF0);  // (narrow maps: past the last column -> out of range)
    }
  }
  const int oplane4 = __builtin_amdgcn_readfirstlane(oplane * ESD);  // bytes of one channel plane of dst
  // NARROW: maps less than one tile wide (16x16, 8x8): lanes past the last column store nothing (their offset is out
  // of the descriptor's range) and count as zeros in the statistics
  const bool lane_ok = ox0 + l31 < p.wout;
#ifdef DSG_H2_TIMING
  unsigned long long rt_e[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (128-cout tiles mark four slabs; the record keeps the first two: its "stats tail" then holds slabs 2, 3)
#define DSG_ET(i) do { __builtin_amdgcn_sched_barrier(0); rt_e[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DSG_ET(i)
#endif
  DSG_ET(0);
  // ONE copy of the epilogue, the statistics under a (uniform) run-time branch and the narrow-map mask always applied:
  // with four compile-time variants behind a four-way branch the compiler hoisted their common head -- the read-out of
  // half the accumulators -- above the branch, spilled 32 of them there and reloaded them in every variant behind the
  // first slab's stores.
  const float ep_scale = (SC && sc_first) ? 1.f : rg_out;  // (shortcut-first tiles took the guard's factor back out already)
  auto epilogue = [&](const bool STATS) {
    constexpr bool NARROW = true;
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) {
      float rv[16][NT], addv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int crel = mt * 32 + (r & 3) + 8 * (r >> 2);  // this lane's channel is crel + 4*half
        addv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bias_rs, 16 * half, crel * 4, 0)) +
                  __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(temb_rs, 16 * half, crel * 4, 0));
        if constexpr (SC)
          addv[r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(scb_rs, 16 * half, crel * 4, 0));
      }
      float gsc[GNB ? 16 : 1], gsh[GNB ? 16 : 1];  // GNB: the norm's (scale, shift) of this lane's 16 channels of the slab
      if constexpr (GNB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          const int crel = mt * 32 + (r & 3) + 8 * (r >> 2);
          const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(gss_rs, 32 * half, crel * 8, 0);
          // (through scalars: hipcc 7.2's front end evaluates __builtin_bit_cast(float, q.y) on the vector's FIRST element --
          //  the element lvalue decays to the vector's address -- and the load is then narrowed to one dword: profiles/FINDINGS.md)
          const unsigned q0 = q.x, q1 = q.y;
          gsc[r] = __builtin_bit_cast(float, q0);
          gsh[r] = __builtin_bit_cast(float, q1);
        }
      }
      const __amdgpu_buffer_rsrc_t res_rs = (GNB && mt * 32 >= gnb_seam) ? res_rs1 : res_rs0;  // (uniform)
      if (has_r) {
        if constexpr (DB) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              if constexpr (ESD == 2) {  // four 16-bit channels: 8 bytes
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(res_rs, voff[nt], (mt * 4 + rg) * 8 * oplane4, 0);
                rv[4 * rg][nt] = lo16<PREC>(q.x); rv[4 * rg + 1][nt] = hi16<PREC>(q.x);
                rv[4 * rg + 2][nt] = lo16<PREC>(q.y); rv[4 * rg + 3][nt] = hi16<PREC>(q.y);
              } else {
                const float4 q = __builtin_bit_cast(
                    float4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, voff[nt], (mt * 4 + rg) * 8 * oplane4, 0));
                rv[4 * rg][nt] = q.x; rv[4 * rg + 1][nt] = q.y; rv[4 * rg + 2][nt] = q.z; rv[4 * rg + 3][nt] = q.w;
              }
            }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int crel = mt * 32 + (r & 3) + 8 * (r >> 2);
              rv[r][nt] = __builtin_bit_cast(
                  float, __builtin_amdgcn_raw_buffer_load_b32(res_rs, voff[nt], crel * oplane4, 0));
            }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) rv[r][nt] = 0.f;
      }
#ifdef DSG_H2_TIMING
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (timing build: the slab's loads have landed)
#endif
      DSG_ET(1 + 2 * mt);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {  // a register group = four consecutive output channels
        float vv[4][NT];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int r = 4 * rg + j;
            const float radd = GNB ? 0.f : rv[r][nt];
            if constexpr (NP == 2 && (ACT != 2 || SC))
              vv[j][nt] = ((acc_hi[mt][nt][r] + acc_lo[mt][nt][r] * (1.0f / 2048.0f)) * ep_scale + addv[r]) + radd;
            else if constexpr (NP == 2)
              vv[j][nt] = ((acc_hi[mt][nt][r] + acc_lo[mt][nt][r] * (1.0f / 2048.0f)) + addv[r]) + radd;
            else
              vv[j][nt] = (acc_hi[mt][nt][r] + addv[r]) + radd;
          }
        if constexpr (DB) {  // the group is 16 contiguous bytes of the pixel's channel block
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const float4 o = make_float4(vv[0][nt], vv[1][nt], vv[2][nt], vv[3][nt]);
            // The channel-block offset goes into the VECTOR offset on purpose.  With an SGPR soffset hipcc 7.2 treats
            // the 16-byte store as free of the "VALU overwrites store data" hazard and re-uses the data registers
            // two or three instructions later; on gfx950 that corrupted the second dword of lanes 12..15 of every
            // row (found by the bit-exact layout tests).  Without an soffset register it inserts the wait states.
#ifndef DSG_H2_TIMING_NOSTORE
            if constexpr (ESD == 2) {  // rounded to the 16-bit type: the group is 8 bytes
              typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
              const u32x2 o2 = {pack2<PREC>(o.x, o.y), pack2<PREC>(o.z, o.w)};
              __builtin_amdgcn_raw_buffer_store_b64(o2, dst_rs, voff[nt] + (mt * 4 + rg) * 8 * oplane4, 0, 0);
            } else {
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), dst_rs,
                                                     voff[nt] + (mt * 4 + rg) * 8 * oplane4, 0, 0);
            }
#else
            if (o.x == 1234.5f) red[lane] = o.y + o.z + o.w;  // (timing experiment: keep the math, drop the stores)
#endif
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vv[j][nt]), dst_rs, voff[nt],
                                                    (mt * 32 + j + 8 * rg) * oplane4, 0);
        }
        if (STATS) {  // GroupNorm statistics of the tensor just produced (the next layer's norm reads them)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int pr = 0; pr < NT / 2; ++pr) {  // one partial per pair of rows: the same summation tree for any NT
              float a = (NARROW && !lane_ok) ? 0.f : vv[j][2 * pr], b = (NARROW && !lane_ok) ? 0.f : vv[j][2 * pr + 1];
              const int v = ((rg * 4 + j) * (NT / 2) + pr) * 2;
              if constexpr (GNB) {  // (sum du, sum du * x) of the row pair, du = dA * silu'(x * sc + sh), dA as it is STORED
                const int r = 4 * rg + j;
                const float xa = rv[r][2 * pr], xb = rv[r][2 * pr + 1];
                if constexpr (ESD == 2) {
                  a = lo16<PREC>(pack2<PREC>(a, 0.f));
                  b = lo16<PREC>(pack2<PREC>(b, 0.f));
                }
                if (p.gnb_silu) {
                  a *= dsilu_fast(__builtin_fmaf(xa, gsc[r], gsh[r]));
                  b *= dsilu_fast(__builtin_fmaf(xb, gsc[r], gsh[r]));
                }
                stab[stab_sw[v & 3] + v * 64] = a + b;
                stab[stab_sw[(v + 1) & 3] + (v + 1) * 64] = __builtin_fmaf(a, xa, b * xb);
              } else {
                stab[stab_sw[v & 3] + v * 64] = a + b;
                stab[stab_sw[(v + 1) & 3] + (v + 1) * 64] = __builtin_fmaf(a, a, b * b);  // (explicit: every instantiation must round alike)
              }
            }
          }
        }
      }
      if (STATS) {
        // table row = (value, half-wave): 32 consecutive floats; same-wave LDS operations complete in order, so the
        // rows are there when they are read
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rr = 0; rr < (2 * SV + 63) / 64; ++rr) {
          const int row = lane + 64 * rr;
          if (2 * SV >= 64 * (rr + 1) || row < 2 * SV) {
            const int v = row >> 1, hh = row & 1;
            const float4* rp = reinterpret_cast<const float4*>(stab + row * 32);
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float4 q4 = rp[k ^ (row & 7)];
              t = (((t + q4.x) + q4.y) + q4.z) + q4.w;
            }
            const int which = v & 1, pr = (v >> 1) % (NT / 2), cj = (v >> 1) / (NT / 2);  // cj = rg * 4 + j
            const int crel = mt * 32 + (cj & 3) + 8 * (cj >> 2);
            red[((wave * (NT / 2) + pr) * 2 + which) * BM + crel + 4 * hh] = t;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      DSG_ET(2 + 2 * mt);
    }
  };
  epilogue(want_stats);
  if (want_stats) {
    // statistics tiles are 8 rows x 32 columns (4 row pairs, summed in row order in fp64) whatever NT is, so the
    // values -- and everything downstream of the norm -- do not depend on the launch geometry
    __syncthreads();
    if (tid < 2 * BM) {
      const int cl = tid & (BM - 1), which = (tid / BM) & 1;
      if (m0 + cl < p.cout) {
        constexpr int NE = NW * NT / 8;  // 8-row statistics tiles per workgroup tile
        const int ntile1 = p.tiles_x * p.tiles_y * NE;           // entries per phase
        const int ntile = ntile1 * (GM == 2 ? 4 : 1);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          double t = 0.0;
#pragma unroll
          for (int j = 0; j < 4; ++j) t += (double)red[((4 * e + j) * 2 + which) * BM + cl];
          const int tile8 = phase * ntile1 + (ty * NE + e) * p.tiles_x + tx;
#ifndef DSG_H2_TIMING
          p.stats[(((size_t)n * p.cout + m0 + cl) * ntile + tile8) * 2 + which] = t;
#endif
        }
      }
    }
  }
#ifdef DSG_H2_TIMING
  // one 32-double record per workgroup, written by its first lane only (no atomics: 30k waves adding into one word
  // were most of the instrumented kernel's time); tools/h2_timing.py reads them
  if (p.stats && tid == 0) {
    double* rec = p.stats + 32 * (size_t)blockIdx.x;
    rec[0] = (double)rt_entry; rec[1] = (double)rt_loop; rec[2] = (double)rt_loop_end; rec[3] = (double)t_loop_cycles;
    rec[4] = (double)t_vm; rec[5] = (double)t_bar;
    for (int k = 0; k < 4; ++k) rec[6 + k] = (double)rt_p[k];
    rt_e[5] = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < 6; ++k) rec[10 + k] = (double)rt_e[k];
    for (int t = 0; t < 10; ++t) rec[17 + t] = (double)t_tap[t];
    rec[16] = (double)__builtin_amdgcn_s_memrealtime();
  }
#endif
}

}  // namespace dsg
