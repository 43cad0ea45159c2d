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
// (tests/test_gpu_ops.py::test_conv_h2_*).  Inputs must satisfy |x| < 65504 (GroupNorm/SiLU outputs and
// residual-stream activations do).
//
// LDS images (per K-chunk of 16 channels, double-buffered; same bytes as the fp32 kernel's):
//   X[piece 2][g 2][pos 10x34][8 halfs]   -- lane = pixel reads one 16-B fragment (k-group g = lane>>5)
//   W[piece 2][tap 9][g 2][cout 64][8]    -- lane = cout  reads one 16-B fragment; filled by LDS-DMA
//                                            (global_load_lds_dwordx4: the pre-split weights need no math)
// A and B use the same (g, j) <-> channel 8g+j map, so the MFMA's internal k order is irrelevant.
#include "dsg_common.h"
#include <algorithm>
#include <type_traits>

namespace dsg {

bool prof_on();
int prof_begin(int kid, double flops, double bytes, hipStream_t st);
void prof_end(int idx, hipStream_t st);

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct ConvH2P {
  const float* src0;
  const float* src1;
  int c0, c1, cin;
  int n, hin, win;
  int hc, wc;
  int hout, wout;
  int cout, cout_pad;
  const _Float16* wh;  // [cin/16][2][9][2][cout_pad][8]
  const float* bias;
  const float* ss;
  int silu;
  const float* temb;
  int temb_stride;
  const float* res;
  float* dst;
  double* stats;  // optional [n][cout][hout/8 * wout/32][2]: per-tile (sum, sum of squares) of the values written
  int tiles_x, tiles_y;
};

constexpr int H2_TW = 32, H2_KC = 16, H2_BM = 64;

// NT = output rows per wave (2 or 4): a workgroup covers 4*NT rows x 32 cols.  NT = 4 halves the LDS operand
// traffic per MFMA (each weight fragment feeds 4 pixel tiles) and the weight DMA per MFMA; it needs 256
// accumulator registers (the kernel owns the SIMD: 1 wave, 512 registers).
// KS = 3 (halo of 1) or 1 (no halo; attention projections and resnet shortcuts)
template <int NT, int KS>
struct H2Geom {
  static constexpr int TAPS = KS * KS;
  static constexpr int TH = 4 * NT;
  static constexpr int PH = TH + KS - 1;
  static constexpr int PW = H2_TW + KS - 1;
  static constexpr int PSZ = PW * PH;                 // KS=3: 340 (NT=2) / 612 (NT=4); KS=1: 256 / 512
  static constexpr int WHALFS = 2 * TAPS * 2 * H2_BM * 8;  // [piece][tap][g][cout][8]: 36864 B / 4096 B
  static constexpr int XHALFS = 2 * 2 * PSZ * 8;      // [piece][g][pos][8]
  static constexpr int BUF_BYTES = (WHALFS + XHALFS) * 2 + 64;  // + a dump slot for masked lanes
  static constexpr int FULL = PSZ / 256;              // full 256-position slabs per k-group
  static constexpr bool HAS_REM = (PSZ % 256) != 0;   // KS=3 leaves a remainder slab shared by the two k-groups
  static constexpr int NU = 2 * FULL + (HAS_REM ? 1 : 0);  // staging units per thread
  static constexpr int REM0 = FULL * 256;             // first position of the remainder unit
  static constexpr int NSEG = 4 * TAPS;               // 1-KB weight segments per chunk
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

__device__ __forceinline__ float silu_fast_h(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// GM: 0 plain, 1 nearest x2 gather; NT: rows per wave; KS: 3 | 1.
// Staging units are arranged so that the k-group g (hence the channel plane and the GroupNorm scale/shift) of
// every unit is WAVE-UNIFORM: channel-plane bases and scale/shift live in SGPRs (s_load / saddr-form global
// loads), and the only per-lane address is the 32-bit halo offset computed once per tile.
//   units 0..FULL-1: g = 0, halo positions tid + 256*i      units FULL..2*FULL-1: g = 1, same positions
//   last unit: g = wave >> 1, halo position FULL*256 + (tid & 127)   (the remainder, valid where < PSZ)
template <int GM, int NT, int KS, int EXP = 0>
__global__ __launch_bounds__(256, 1) void conv_h2_kernel(ConvH2P p) {
  using G = H2Geom<NT, KS>;
  constexpr int H2_TH = G::TH, H2_PSZ = G::PSZ, H2_XHALFS = G::XHALFS, H2_BUF_BYTES = G::BUF_BYTES, H2_NU = G::NU;
  constexpr int FULL = G::FULL, TAPS = G::TAPS, H2_PW = G::PW, H2_WHALFS = G::WHALFS, PADK = KS / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;

  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int m0 = blockIdx.y * H2_BM;
  const int oy0 = ty * H2_TH, ox0 = tx * H2_TW;
  const int plane = p.hin * p.win;
  const int nq = p.cin / H2_KC;
  const int g2 = wave >> 1;  // k-group of unit 2 (uniform per wave)

  int goff[H2_NU], xoff[H2_NU];
  unsigned valid = 0;
#pragma unroll
  for (int i = 0; i < H2_NU; ++i) {
    const int g = i < FULL ? 0 : (i < 2 * FULL ? 1 : g2);
    const int pos = i < 2 * FULL ? tid + 256 * (i % FULL) : G::REM0 + (tid & 127);
    int off = 0, xo = H2_WHALFS + H2_XHALFS;  // dump slot (in halfs) when the position is past the patch
    if (pos < H2_PSZ) {
      const int py = pos / H2_PW, px = pos - py * H2_PW;
      const int gy = oy0 - PADK + py, gx = ox0 - PADK + px;
      if (gy >= 0 && gy < p.hc && gx >= 0 && gx < p.wc) {
        off = (GM ? (gy >> 1) : gy) * p.win + (GM ? (gx >> 1) : gx);
        valid |= 1u << i;
      }
      xo = H2_WHALFS + (g * H2_PSZ + pos) * 8;  // piece 0; piece 1 is 2*PSZ*8 halfs further
    }
    goff[i] = off;
    xoff[i] = xo;
  }
  const bool has_ss = p.ss != nullptr;
  const bool do_silu = has_ss && p.silu;
  const float* ssg = has_ss ? p.ss + (size_t)n * p.cin * 2 : nullptr;

  float xr[H2_NU][8];
  float2 sr[H2_NU][8];  // GroupNorm (scale, shift) of the unit's 8 channels, fetched with the patch (one chunk ahead)

  auto src_of = [&](int q) -> const float* {  // uniform
    const int cb = q * H2_KC;
    return (cb < p.c0) ? p.src0 + ((size_t)n * p.c0 + cb) * plane
                       : p.src1 + ((size_t)n * p.c1 + (cb - p.c0)) * plane;
  };
  auto unit_g = [&](int i) -> int { return i < FULL ? 0 : (i < 2 * FULL ? 1 : g2); };
  auto load_unit = [&](int i, int q, const float* sp) {
    const float* spg = sp + (size_t)(unit_g(i) * 8) * plane;  // uniform
#pragma unroll
    for (int j = 0; j < 8; ++j) xr[i][j] = (spg + (size_t)j * plane)[goff[i]];
    if (has_ss) {
      const float* ssq = ssg + 2 * (q * H2_KC + unit_g(i) * 8);  // uniform address: one line, broadcast
#pragma unroll
      for (int j = 0; j < 8; ++j) sr[i][j] = *reinterpret_cast<const float2*>(ssq + 2 * j);
    }
  };
  auto commit_unit = [&](int i, unsigned char* buf) {
    half8 h1, h2;
    const bool ok = (valid >> i) & 1u;
    if (EXP == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) h1[j] = (_Float16)xr[i][j];
      _Float16* xb = reinterpret_cast<_Float16*>(buf);
      *reinterpret_cast<half8*>(xb + xoff[i]) = h1;
      *reinterpret_cast<half8*>(xb + xoff[i] + (xoff[i] < H2_WHALFS + H2_XHALFS ? 2 * H2_PSZ * 8 : 0)) = h1;
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = xr[i][j];
      if (has_ss) v = v * sr[i][j].x + sr[i][j].y;
      const float sv = silu_fast_h(v);
      v = do_silu ? sv : v;
      v = ok ? v : 0.f;
      const _Float16 a = (_Float16)v;
      h1[j] = a;
      h2[j] = (_Float16)((v - (float)a) * 2048.0f);
    }
    _Float16* xb = reinterpret_cast<_Float16*>(buf);
    *reinterpret_cast<half8*>(xb + xoff[i]) = h1;
    *reinterpret_cast<half8*>(xb + xoff[i] + (xoff[i] < H2_WHALFS + H2_XHALFS ? 2 * H2_PSZ * 8 : 0)) = h2;
  };
  // weight slab of chunk q: 36 segments (piece, tap, g) of 64 couts x 16 B, moved global -> LDS by DMA;
  // wave w moves segments w, w+4, ...
  auto dma_weights = [&](int k, int q, unsigned char* buf) {
    const int seg = wave + 4 * k;  // 0..NSEG-1
    const _Float16* gp = p.wh + (((size_t)q * G::NSEG + seg) * p.cout_pad + m0) * 8;  // uniform
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + lane * 8),
                                     (__attribute__((address_space(3))) void*)(buf + seg * 1024), 16, 0, 0);
  };

  f32x16 acc_hi[2][NT], acc_lo[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc_hi[mt][nt][r] = 0.f;
        acc_lo[mt][nt][r] = 0.f;
      }

  unsigned char* buf0 = smem_raw;
  unsigned char* buf1 = smem_raw + H2_BUF_BYTES;

  // prologue: chunk 0 -> buffer 0; chunk 1 -> registers
  {
    const float* sp = src_of(0);
#pragma unroll
    for (int i = 0; i < H2_NU; ++i) load_unit(i, 0, sp);
#pragma unroll
    for (int k = 0; k < TAPS; ++k) dma_weights(k, 0, buf0);
#pragma unroll
    for (int i = 0; i < H2_NU; ++i) commit_unit(i, buf0);
    if (nq > 1) {
      const float* sp1 = src_of(1);
#pragma unroll
      for (int i = 0; i < H2_NU; ++i) load_unit(i, 1, sp1);
    }
  }
  __syncthreads();

  // One K-chunk: MFMAs on `cur`; STAGE: chunk q+1 (patch in registers, weights by DMA) goes into `nxt`;
  // LOAD: chunk q+2's patch is fetched into the registers just freed.
  auto chunk = [&](int q, auto stage_tag, auto load_tag) {
    constexpr bool STAGE = decltype(stage_tag)::value, LOAD = decltype(load_tag)::value;
    unsigned char* cur = (q & 1) ? buf1 : buf0;
    unsigned char* nxt = (q & 1) ? buf0 : buf1;
    const float* spn = LOAD ? src_of(q + 2) : nullptr;
    const _Float16* wl = reinterpret_cast<const _Float16*>(cur);
    const _Float16* xl = wl + H2_WHALFS;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      if (KS == 1) {  // one tap: all units and the four weight segments ride on it
#pragma unroll
        for (int u = 0; u < H2_NU; ++u) {
          if (STAGE) commit_unit(u, nxt);
          if (LOAD) load_unit(u, q + 2, spn);
        }
        if (STAGE) dma_weights(0, q + 1, nxt);
      }
      // KS = 3 -- NU = 3: units at taps 1, 4, 7;  NU = 5: units at taps 0, 2, 4, 6, 8
      constexpr int UNIT_STRIDE = (H2_NU == 3) ? 3 : 2, UNIT_PHASE = (H2_NU == 3) ? 1 : 0;
      if (EXP < 2 && KS == 3 && tap % UNIT_STRIDE == UNIT_PHASE && tap / UNIT_STRIDE < H2_NU) {
        if (STAGE) commit_unit(tap / UNIT_STRIDE, nxt);
        if (LOAD) load_unit(tap / UNIT_STRIDE, q + 2, spn);
      }
      // The weight DMAs go AFTER this tap's commit: with a DMA in flight hipcc waits vmcnt(0) at every use of
      // an ordinary load result, so a DMA issued just before a commit would stall it for the whole transfer.
      if (STAGE && KS == 3) {
        if (H2_NU == 5) {  // commits at even taps: two DMAs at each odd tap, the ninth after the last commit
          if (tap & 1) {
            dma_weights(tap - 1, q + 1, nxt);
            dma_weights(tap, q + 1, nxt);
          } else if (tap == 8) {
            dma_weights(8, q + 1, nxt);
          }
        } else {  // commits at taps 1, 4, 7: three DMAs right after each
          if (tap % 3 == 1) {
            dma_weights(tap - 1, q + 1, nxt);
            dma_weights(tap, q + 1, nxt);
            dma_weights(tap + 1, q + 1, nxt);
          }
        }
      }
      const int dy = tap / KS, dx = tap % KS;
      half8 a[2][2], b[NT][2];  // [tile][piece]
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
          a[mt][pc] = *reinterpret_cast<const half8*>(wl + (((pc * TAPS + tap) * 2 + half) * H2_BM + mt * 32 + l31) * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
          b[nt][pc] = *reinterpret_cast<const half8*>(
              xl + ((pc * 2 + half) * H2_PSZ + (wave * NT + nt + (EXP == 3 ? 0 : dy)) * H2_PW + l31 + (EXP == 3 ? 0 : dx)) * 8);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc_hi[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][0], b[nt][0], acc_hi[mt][nt], 0, 0, 0);
          acc_lo[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][0], b[nt][1], acc_lo[mt][nt], 0, 0, 0);
          acc_lo[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][1], b[nt][0], acc_lo[mt][nt], 0, 0, 0);
        }
    }
    __syncthreads();  // (drains the DMA: nxt is complete; everyone is done reading cur)
  };
  using T = std::true_type;
  using F = std::false_type;
  int q = 0;
  for (; q + 2 < nq; ++q) chunk(q, T{}, T{});
  if (q + 1 < nq) chunk(q++, T{}, F{});  // last staged chunk: nothing left to load
  chunk(q, F{}, F{});                    // last chunk: MFMAs only

  // Epilogue.  Everything but the lane's (half, column) offset is wave-uniform, so row bases, bias and
  // time-embedding values come through SGPRs; all residual loads are issued before the first use (with one
  // wave per SIMD a load->add->store chain per element would expose the memory latency 64 times).
  // (the host only dispatches here when cout % 8 == 0, so a 4-row half-group is never split by cout)
  const bool has_t = p.temb != nullptr;
  const bool has_r = p.res != nullptr;
  const int oplane = p.hout * p.wout;
  const int lane_off = 4 * half * oplane + l31;
  const bool want_stats = p.stats != nullptr;
  float* red = reinterpret_cast<float*>(smem_raw);  // [wave][sum | sumsq][cout 64] (the K loop is done with LDS)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    float rv[16][NT];  // residual values of this 32-cout slab, all in flight before the first use
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cou = min(m0 + mt * 32 + (r & 3) + 8 * (r >> 2), p.cout - 8 + (r & 3));  // uniform, in range
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float* row = p.res + (((size_t)n * p.cout + cou) * p.hout + oy0 + wave * NT + nt) * p.wout + ox0;
        rv[r][nt] = has_r ? row[lane_off] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cou = m0 + mt * 32 + (r & 3) + 8 * (r >> 2);  // uniform; this lane's cout is cou + 4*half
      if (cou < p.cout) {
        float add0 = 0.f, add1 = 0.f;
        if (p.bias) {
          add0 = p.bias[cou];
          add1 = p.bias[cou + 4];
        }
        if (has_t) {
          add0 += p.temb[(size_t)n * p.temb_stride + cou];
          add1 += p.temb[(size_t)n * p.temb_stride + cou + 4];
        }
        const float add = half ? add1 : add0;
        float s1[NT / 2], s2[NT / 2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float* row = p.dst + (((size_t)n * p.cout + cou) * p.hout + oy0 + wave * NT + nt) * p.wout + ox0;
          float v = (acc_hi[mt][nt][r] + acc_lo[mt][nt][r] * (1.0f / 2048.0f)) + add;
          if (has_r) v = v + rv[r][nt];
          row[lane_off] = v;
          if (nt & 1) {
            s1[nt / 2] += v;
            s2[nt / 2] += v * v;
          } else {
            s1[nt / 2] = v;
            s2[nt / 2] = v * v;
          }
        }
        if (want_stats) {  // GroupNorm statistics of the tensor just produced (the next layer's norm reads them)
          const int cl = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
          for (int pr = 0; pr < NT / 2; ++pr) {  // one partial per pair of rows: the same summation tree for any NT
            const float t1 = half_wave_sum(s1[pr]), t2 = half_wave_sum(s2[pr]);
            if (l31 == 16) {
              red[((wave * (NT / 2) + pr) * 2 + 0) * H2_BM + cl] = t1;
              red[((wave * (NT / 2) + pr) * 2 + 1) * H2_BM + cl] = t2;
            }
          }
        }
      }
    }
  }
  if (want_stats) {
    // statistics tiles are 8 rows x 32 columns (4 row pairs, summed in row order in fp64) whatever NT is, so the
    // values -- and everything downstream of the norm -- do not depend on the launch geometry
    __syncthreads();
    if (tid < 2 * H2_BM) {
      const int cl = tid & (H2_BM - 1), which = tid >> 6;
      if (m0 + cl < p.cout) {
        const int ntile = p.tiles_x * p.tiles_y * (NT / 2);
#pragma unroll
        for (int e = 0; e < NT / 2; ++e) {
          double t = 0.0;
#pragma unroll
          for (int j = 0; j < 4; ++j) t += (double)red[((4 * e + j) * 2 + which) * H2_BM + cl];
          const int tile8 = (ty * (NT / 2) + e) * p.tiles_x + tx;
          p.stats[(((size_t)n * p.cout + m0 + cl) * ntile + tile8) * 2 + which] = t;
        }
      }
    }
  }
}

// OIHW fp32 -> [cin/16][piece 2][tap k*k][g 2][cout_pad][8] fp16 (hi, scaled lo); zero-padded couts.
// mode 1 (data gradient): the conv dX = conv(dY, W^T flipped) has K = cout, N = cin: the same layout with the
// roles of the two channel axes swapped and the taps reversed.  `cout_off` places this weight's N columns inside a
// wider matrix (fused q/k/v projection).
__global__ void weight_relayout_h2_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int cout_w, int cin_w,
                                          int taps, int cout_pad, int cout_off, int mode) {
  const int cin = mode ? cout_w : cin_w;   // K axis of the conv this layout feeds
  const int cout = mode ? cin_w : cout_w;  // N axis (this weight's share of it)
  const int64_t total = (int64_t)(cin / 16) * taps * 2 * cout * 8;  // one thread per (chunk, tap, g, co, j)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % 8);
    int64_t r = i / 8;
    const int co = (int)(r % cout);
    r /= cout;
    const int g = (int)(r % 2);
    r /= 2;
    const int tap = (int)(r % taps);
    const int q = (int)(r / taps);
    const int ci = q * 16 + g * 8 + j;
    const float v = mode ? w[((int64_t)ci * cin_w + co) * taps + (taps - 1 - tap)] : w[((int64_t)co * cin_w + ci) * taps + tap];
    const _Float16 h1 = (_Float16)v;
    const _Float16 h2 = (_Float16)((v - (float)h1) * 2048.0f);
    const int64_t base = ((((int64_t)q * 2 + 0) * taps + tap) * 2 + g) * cout_pad + cout_off + co;
    const int64_t base1 = ((((int64_t)q * 2 + 1) * taps + tap) * 2 + g) * cout_pad + cout_off + co;
    dst[base * 8 + j] = h1;
    dst[base1 * 8 + j] = h2;
  }
}

static int g_h2_enabled = 1;
static int g_h2_stats = 1;  // epilogue GroupNorm statistics (tuning key 5: A/B against the separate pass)
static int g_h2_exp = 0;  // experiment variants (tools/ only)
static int g_h2_rows = 0;  // rows per wave: 0 = by grid size, 2 | 4 forced (tuning key 3)

bool conv_h2_eligible(const dsg_conv_args* a, int hout, int wout) {
  const int cin = a->c0 + a->c1;
  if (a->ksize == 1)  // pointwise: the map is re-tiled as (h*w/32) rows of 32 pixels, so only h*w matters
    return g_h2_enabled && a->weight_h2 != nullptr && a->stride == 1 && !a->upsample && !a->pool2 && !a->temb &&
           cin % 16 == 0 && (a->c1 == 0 || a->c0 % 16 == 0) && (hout * wout) % (8 * H2_TW) == 0 && a->cout % 8 == 0;
  return g_h2_enabled && a->weight_h2 != nullptr && a->stride == 1 && a->upsample <= 1 && !a->pool2 &&
         cin % 16 == 0 && (a->c1 == 0 || a->c0 % 16 == 0) && (wout % H2_TW == 0) && (hout % 8 == 0) &&
         a->cout % 8 == 0;
}

// tile geometry shared by the launcher and dsg_conv2d_stats_tiles
static bool conv_h2_rows16(const dsg_conv_args* a, int hout, int wout) {
  if (a->ksize == 1) {
    hout = hout * wout / H2_TW;
    wout = H2_TW;
  }
  const int cout_pad = (a->cout + 63) / 64 * 64;
  const int blocks16 = (hout % 16 == 0) ? (wout / H2_TW) * (hout / 16) * a->n * (cout_pad / H2_BM) : 0;
  return g_h2_rows != 2 && blocks16 >= (g_h2_rows == 4 ? 1 : 256);
}

int conv_h2_stats_tiles(const dsg_conv_args* a, int hout, int wout) {
  if (!g_h2_stats || !conv_h2_eligible(a, hout, wout)) return 0;
  return hout * wout / (H2_TW * 8);  // 8-row x 32-column statistics tiles for either block height
}

int conv_h2_launch(const dsg_conv_args* a, int hout, int wout, hipStream_t st) {
  const bool nt4 = conv_h2_rows16(a, hout, wout);
  ConvH2P p;
  p.stats = a->stats_out;
  p.src0 = a->src0; p.src1 = a->src1; p.c0 = a->c0; p.c1 = a->c1; p.cin = a->c0 + a->c1;
  p.n = a->n; p.hin = a->hin; p.win = a->win;
  if (a->ksize == 1) {
    hout = hout * wout / H2_TW; wout = H2_TW;
    p.hin = hout; p.win = wout;
  }
  p.hc = a->upsample ? 2 * p.hin : p.hin;
  p.wc = a->upsample ? 2 * p.win : p.win;
  p.hout = hout; p.wout = wout; p.cout = a->cout; p.cout_pad = (a->cout + 63) / 64 * 64;
  p.wh = static_cast<const _Float16*>(a->weight_h2);
  p.bias = a->bias; p.ss = a->gn_scale_shift; p.silu = a->silu; p.temb = a->temb; p.temb_stride = a->temb_stride;
  p.res = a->residual; p.dst = a->dst;
  // 16-row tiles (NT = 4) when they still give every CU a workgroup; 8-row tiles otherwise
  const int th = nt4 ? 16 : 8;
  p.tiles_x = wout / H2_TW; p.tiles_y = hout / th;
  const bool k1 = a->ksize == 1;
  const size_t lds = 2 * (size_t)(k1 ? (nt4 ? H2Geom<4, 1>::BUF_BYTES : H2Geom<2, 1>::BUF_BYTES)
                                     : (nt4 ? H2Geom<4, 3>::BUF_BYTES : H2Geom<2, 3>::BUF_BYTES));
  dim3 grid(p.tiles_x * p.tiles_y * p.n, p.cout_pad / H2_BM);
  int pi = -1;
  if (prof_on()) {
    const double px = (double)p.n * hout * wout;
    const int taps = a->ksize * a->ksize;
    pi = prof_begin(a->ksize == 1 ? 8 : (a->upsample ? 7 : 6), 2.0 * px * p.cout * p.cin * taps,
                    4.0 * ((double)p.n * p.cin * p.hin * p.win + (double)p.cin * taps * p.cout +
                           px * p.cout * (p.res ? 2.0 : 1.0)), st);
  }
  static bool raised = false;
  if (!raised) {
    const void* ks[9] = {reinterpret_cast<const void*>(conv_h2_kernel<0, 4, 3, 1>),
                         reinterpret_cast<const void*>(conv_h2_kernel<0, 4, 3, 2>),
                         reinterpret_cast<const void*>(conv_h2_kernel<0, 4, 3, 3>),
                         reinterpret_cast<const void*>(conv_h2_kernel<0, 2, 3>), reinterpret_cast<const void*>(conv_h2_kernel<1, 2, 3>),
                         reinterpret_cast<const void*>(conv_h2_kernel<0, 4, 3>), reinterpret_cast<const void*>(conv_h2_kernel<1, 4, 3>),
                         reinterpret_cast<const void*>(conv_h2_kernel<0, 2, 1>), reinterpret_cast<const void*>(conv_h2_kernel<0, 4, 1>)};
    for (const void* k : ks) DSG_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised = true;
  }
  if (k1) {
    if (nt4) hipLaunchKernelGGL((conv_h2_kernel<0, 4, 1>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv_h2_kernel<0, 2, 1>), grid, dim3(256), lds, st, p);
  } else if (a->upsample) {
    if (nt4) hipLaunchKernelGGL((conv_h2_kernel<1, 4, 3>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv_h2_kernel<1, 2, 3>), grid, dim3(256), lds, st, p);
  } else {
    if (nt4 && g_h2_exp == 1) hipLaunchKernelGGL((conv_h2_kernel<0, 4, 3, 1>), grid, dim3(256), lds, st, p);
    else if (nt4 && g_h2_exp == 2) hipLaunchKernelGGL((conv_h2_kernel<0, 4, 3, 2>), grid, dim3(256), lds, st, p);
    else if (nt4 && g_h2_exp == 3) hipLaunchKernelGGL((conv_h2_kernel<0, 4, 3, 3>), grid, dim3(256), lds, st, p);
    else if (nt4) hipLaunchKernelGGL((conv_h2_kernel<0, 4, 3>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv_h2_kernel<0, 2, 3>), grid, dim3(256), lds, st, p);
  }
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

void conv_h2_set_enabled(int on) { g_h2_enabled = on; }
void conv_h2_set_rows(int r) { g_h2_rows = r; }
void conv_h2_set_exp(int e) { g_h2_exp = e; }
void conv_h2_set_stats(int on) { g_h2_stats = on; }

}  // namespace dsg

static int relayout_h2(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, int32_t ksize, int32_t n_total,
                       int32_t n_off, int mode, void* stream) {
  DSG_CHECK_ARG(w_oihw && dst_half, "dsg_conv_weight_relayout_h2: NULL pointer");
  DSG_CHECK_ARG(ksize == 1 || ksize == 3, "dsg_conv_weight_relayout_h2: ksize must be 1 or 3");
  const int k_axis = mode ? cout : cin, n_axis = mode ? cin : cout;
  DSG_CHECK_ARG(cout > 0 && cin > 0 && k_axis % 16 == 0,
                "dsg_conv_weight_relayout_h2: the contraction axis (%d) must be a positive multiple of 16", k_axis);
  if (n_total == 0) n_total = n_axis;
  DSG_CHECK_ARG(n_off >= 0 && n_off + n_axis <= n_total, "dsg_conv_weight_relayout_h2: column window out of range");
  const int cout_pad = (n_total + 63) / 64 * 64;
  const int taps = ksize * ksize;
  const int64_t total = (int64_t)(k_axis / 16) * taps * 2 * n_axis * 8;
  const int blocks = (int)std::min<int64_t>(dsg::cdiv64(total, 256), 4096);
  hipLaunchKernelGGL(dsg::weight_relayout_h2_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w_oihw, static_cast<_Float16*>(dst_half), cout, cin, taps, cout_pad, n_off, mode);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_conv_weight_relayout_h2(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, int32_t ksize,
                                        int32_t cout_total, int32_t cout_off, void* stream) {
  return relayout_h2(w_oihw, dst_half, cout, cin, ksize, cout_total, cout_off, 0, stream);
}

DSG_API int dsg_conv_weight_relayout_h2_dgrad(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin,
                                              int32_t ksize, void* stream) {
  return relayout_h2(w_oihw, dst_half, cout, cin, ksize, 0, 0, 1, stream);
}
