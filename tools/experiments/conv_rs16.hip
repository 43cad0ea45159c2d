// Row-streaming, weight-stationary 3x3 convolution for the 64 -> 64 channel layers in the 16-bit modes (bf16 / fp16 channel-blocked
// tensors, one MFMA per product): round 4, VERDICT r03 item 5 ("the 64-channel 256^2 kernel is at 2.82 TB/s (0.35 of HBM) with
// MFMA, VALU and memory time adding up unoverlapped").
//
// Reference call sites: conv1 / conv2 of the 64-channel resnets of diffusers' UNet2DModel as DriveSceneGen builds it
// (DriveSceneGen/scripts/train.py:39-57: down_blocks.0 at 256 x 256), run under accelerate's mixed precision
// (train.py:24 'fp16'; BASELINE configs[4]: bf16) at training_pipeline.py:84-86 (forward and, with ACT = 0 and the PACK_DGRAD
// image, the data gradient) and in the sampling loops (training_pipeline.py:26-32, generation.py:14-20).
//
// The fp32-equivalent form of this kernel (tools/experiments/conv_rs.hip) lost to conv_h2_kernel: with three MFMAs
// per product that layer is matrix-bound and the row-streaming form's LDS round trip for the cross-wave sum came on top.  In 16
// bits the same layer is MEMORY-bound (1.6 GB of tensors against 103 us of matrix time at B=64) and conv_h2_kernel's per-tile
// prologue / epilogue latency chains are what it waits for (560 us = 2.8 TB/s).  Here nothing is per tile:
//   * wave w keeps the weight slice of ITS 16 input channels in registers: 9 taps x 2 cout tiles = 18 fragments = 72 registers,
//     pinned to AGPRs through the asm MFMA, loaded once per workgroup;
//   * the workgroup walks a 32-row band of one 32-column strip; input rows come through a ring of four register sets, loaded
//     three rows ahead (a continuous stream, no per-tile start-up), are normalised / activated / rounded into a wave-private LDS
//     ring of four rows (ACT = 2), or copied as they are (ACT = 0: data-gradient convs);
//   * per output row a wave issues 18 MFMAs against 9 fragment reads, leaves its partial 64 x 32 tile (fp32) in LDS, ONE barrier,
//     and wave w' adds the four partials of its two channel blocks in wave order, adds bias / temb / residual, rounds, stores 8
//     bytes per lane and block and keeps the GroupNorm statistics (of the unrounded values, like conv_h2_kernel) of the 8-row x
//     32-column tile -- one row behind the matrix instructions, every piece dealt to a slot behind one of them.
// Summation order: per wave over its 16 channels x 9 taps in one fp32 accumulator, then ((w0 + w1) + (w2 + w3)): a function of
// the layer only (rows of a batch are bitwise their batch-1 results); not conv_h2_kernel's order -- same rounding class.
#include "conv_h2_launch.h"

namespace dsg {

struct ConvRs16P {
  const unsigned short* src;  // [N][8][H][W][8] 16-bit
  const unsigned short* wh;   // [4][1][9][2][64][8]: PACK_FWD / PACK_DGRAD image of the 64 x 64 x 3 x 3 weight in the 16-bit type
  const float* ss;            // [N][64][2] GroupNorm (scale, shift) (ACT = 2)
  const float* bias;          // [64] or null
  const float* temb;          // [N][temb_stride] or null
  int temb_stride;
  const unsigned short* res;  // like dst, or null
  unsigned short* dst;        // [N][8][H][W][8] 16-bit
  double* stats;              // [N][64][H/8 * W/32][2] or null
  int n, h, w, band;
};

typedef _Float16 r6_half8 __attribute__((ext_vector_type(8)));
typedef float r6_f32x16 __attribute__((ext_vector_type(16)));

constexpr int R6_COLS = 36;                          // ring columns per row (34 in use)
constexpr int R6_SLOTS = 6;                          // input rows in the ring: four in use by a step's two output rows, two being filled
constexpr int R6_ROW_HALFS = 4 * 2 * R6_COLS * 8;    // [chunk 4][k-group 2][column][8 channels]: 4608 B
constexpr int R6_X_BYTES = R6_SLOTS * R6_ROW_HALFS * 2;
constexpr int R6_ST_FLOATS = 2 * 32 * 64;            // statistics hand-over: [cout tile 2][value 32][lane 64]
constexpr int R6_LDS_BYTES = R6_X_BYTES + R6_ST_FLOATS * 4 + 128 * 4;

template <int PREC>
__device__ __forceinline__ void r6_mma(r6_f32x16& c, const r6_half8& a, const r6_half8& b) {
  if constexpr (PREC == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
}
template <int PREC>
__device__ __forceinline__ void r6_mma0(r6_f32x16& c, const r6_half8& a, const r6_half8& b) {  // c = a * b
  if constexpr (PREC == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "a"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "a"(a), "v"(b));
}
__device__ __forceinline__ void r6_pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void r6_pin(unsigned& x) { asm volatile("" : "+v"(x)); }

// Workgroup = one 32-column strip x a band of rows of one image; a STEP is two output rows.  Wave w = (cout tile mt = w & 1, row
// rr = w >> 1) contracts ALL 64 input channels of output row c + rr for its 32 output channels: 4 chunks x 9 taps = 36 MFMAs
// against 36 fragment reads of the shared ring, no cross-wave sum.  Its weight slice (36 fragments = 144 registers) lives in
// AGPRs.  The ring holds six input rows of all 64 channels; every step the workgroup stages two new rows (one 16-byte item per
// thread and row + the halo columns), loaded three steps ahead through a ring of four register sets.  One barrier per step.
// The row finished in step s (bias / temb / residual, rounding, stores, statistics) is dealt out behind the MFMAs of step s + 1.
template <int PREC, int ACT, bool HAS_RES>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_rs16_kernel(ConvRs16P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char r6m[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int mt = wave & 1, rr = wave >> 1;
  unsigned short* xl = reinterpret_cast<unsigned short*>(r6m);
  float* stl = reinterpret_cast<float*>(r6m + R6_X_BYTES);
  float* sst = stl + R6_ST_FLOATS;

  const int tiles_x = p.w / 32, bands = p.h / p.band;
  int id = blockIdx.x;
  const int cx = id % tiles_x;
  id /= tiles_x;
  const int bz = id % bands;
  const int n = id / bands;
  const int r0 = bz * p.band, r1 = r0 + p.band;
  const size_t plane = (size_t)p.h * p.w * 8;  // elements per channel block

  if (ACT == 2 && tid < 128) sst[tid] = p.ss[(size_t)n * 128 + tid];

  // ---- this wave's weight slice: [chunk][tap], 36 fragments (pinned to AGPRs by their only use) -----------------------
  r6_half8 wa[4][9];
  {
    const r6_half8* whp = reinterpret_cast<const r6_half8*>(p.wh);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) wa[q][tap] = whp[(((size_t)q * 9 + tap) * 2 + half) * 64 + mt * 32 + l31];
  }
  __syncthreads();  // (the scale / shift table)

  // ---- staging roles --------------------------------------------------------------------------------------------------
  // main item: thread t = (channel block t >> 5, column t & 31) of a row: 16 bytes; one item per new row and thread
  // halo item: thread t = (row of the step t >> 7, channel block (t >> 4) & 7, side (t >> 3) & 1, channel t & 7): one value
  const int m_blk = tid >> 5, m_col = tid & 31;
  const unsigned short* mrow = p.src + ((size_t)n * 8 + m_blk) * plane + (size_t)(cx * 32 + m_col) * 8;
  const int m_lds = (m_blk * R6_COLS + m_col + 1) * 8;  // [chunk][k-group] = block index; elements inside a slot row
  const int h_row = tid >> 7, h_blk = (tid >> 4) & 7, h_side = (tid >> 3) & 1, h_j = tid & 7;
  const int h_col = cx * 32 + (h_side ? 32 : -1);
  const bool h_colok = (unsigned)h_col < (unsigned)p.w;
  const unsigned short* hrow = p.src + ((size_t)n * 8 + h_blk) * plane + (size_t)(h_colok ? h_col : 0) * 8 + h_j;
  const int h_lds = (h_blk * R6_COLS + (h_side ? 33 : 0)) * 8 + h_j;
  const unsigned short h_keep = h_colok ? 0xFFFF : 0;
  float scv[8], shv[8], h_sc = 1.f, h_sh = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    scv[j] = 1.f;
    shv[j] = 0.f;
  }
  if constexpr (ACT == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      scv[j] = sst[2 * (8 * m_blk + j)];
      shv[j] = sst[2 * (8 * m_blk + j) + 1];
    }
    h_sc = sst[2 * (8 * h_blk + h_j)];
    h_sh = sst[2 * (8 * h_blk + h_j) + 1];
  }
  auto slot_of = [](int i) { return (i + 1 + R6_SLOTS) % R6_SLOTS; };  // ring slot of input row i (i >= -1)
  auto act1 = [&](float x, float sc, float sh) {
    if constexpr (ACT == 2) {
      const float t = x * sc + sh;
      return t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
    } else {
      return x;
    }
  };
  auto word = [](const uint4& q, int k) -> unsigned { return k == 0 ? q.x : (k == 1 ? q.y : (k == 2 ? q.z : q.w)); };
  // input rows: a ring of four register sets of [two rows], loaded three steps ahead of their staging
  uint4 xq[4][2];
  unsigned short hq[4] = {0, 0, 0, 0};
  auto step_load = [&](int i0, int set) {  // rows i0, i0 + 1 (clamped; masked at the commit)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ic = min(max(i0 + k, 0), p.h - 1);
      xq[set][k] = *reinterpret_cast<const uint4*>(mrow + (size_t)ic * p.w * 8);
    }
    const int ih = min(max(i0 + h_row, 0), p.h - 1);
    hq[set] = hrow[(size_t)ih * p.w * 8];
  };
  auto commit_now = [&](int i0, int set) {  // prologue: both rows of a set at once
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = i0 + k;
      const unsigned keep = ((unsigned)i < (unsigned)p.h) ? 0xFFFFFFFFu : 0u;
      unsigned o[4];
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        const unsigned wv = word(xq[set][k], jp);
        if constexpr (ACT == 2)
          o[jp] = pack2<PREC>(act1(lo16<PREC>(wv), scv[2 * jp], shv[2 * jp]), act1(hi16<PREC>(wv), scv[2 * jp + 1], shv[2 * jp + 1])) & keep;
        else
          o[jp] = wv & keep;
      }
      *reinterpret_cast<uint4*>(xl + slot_of(i) * R6_ROW_HALFS + m_lds) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    {
      const int i = i0 + h_row;
      const unsigned short km = (unsigned short)(h_keep & (((unsigned)i < (unsigned)p.h) ? 0xFFFF : 0));
      unsigned short hv = hq[set];
      if constexpr (ACT == 2) hv = (unsigned short)(pack2<PREC>(act1(lo16<PREC>((unsigned)hv), h_sc, h_sh), 0.f) & 0xFFFFu);
      xl[slot_of(i) * R6_ROW_HALFS + h_lds] = (unsigned short)(hv & km);
    }
  };

  // ---- the finishing side: this wave's 32 couts of its row; lane = (pixel l31, half), register r <-> cout mt*32 + 8(r>>2) + 4 half + (r&3)
  float addv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = mt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
    addv[r] = (p.bias ? p.bias[c] : 0.f) + (p.temb ? p.temb[(size_t)n * p.temb_stride + c] : 0.f);
  }
  float s1[16], s2[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) s1[r] = s2[r] = 0.f;
  const size_t obase = ((size_t)n * 8 + mt * 4) * plane + (size_t)(cx * 32 + l31) * 8 + 4 * half;  // + j * plane + row * w * 8
  const int ntile = (p.h / 8) * tiles_x;
  uint2 rres[2][4];  // [ring 2][channel block j]: four 16-bit residual values
#pragma unroll
  for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
    for (int j = 0; j < 4; ++j) rres[a2][j] = make_uint2(0u, 0u);
  auto res_load = [&](int row, int set) {
    if constexpr (HAS_RES) {
      const int rc = min(row, p.h - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) rres[set][j] = *reinterpret_cast<const uint2*>(p.res + obase + j * plane + (size_t)rc * p.w * 8);
    }
  };

  // ---- one step ---------------------------------------------------------------------------------------------------------
  // U = step index & 3 (static).  Output rows c, c + 1 (this wave: c + rr) into acc[U & 1]; finished now: row c - 2 + rr from
  // acc[(U + 1) & 1] with the residual set (U + 1) & 1; staged now: input rows c + 3, c + 4 from register set U; loaded now: rows
  // c + 9, c + 10 into set (U + 3) & 3 and the residual of row c + rr into set U & 1.
  // 66 pieces over 36 slots: 0-15 finishing (per channel block: add, round + store, two statistics pieces), 16-57 the two main
  // items (per value affine + exp, then 1 + e / rcp / mul; per pair the rounding; the 16-byte write), 58-60 the halo value, 61-62 loads.
  r6_f32x16 acc[2];
  const unsigned short* bl = xl + (half * R6_COLS + l31) * 8;  // + slot row, + chunk * 2 * R6_COLS * 8, + dx * 8
  float tv[2][8], ev[2][8], th = 0.f, eh = 0.f, fv[4];
  unsigned ow[2][4];
  auto piece = [&](auto fin_tag, auto u_tag, int q, int c) {
    constexpr bool FIN = decltype(fin_tag)::value;
    constexpr int U = decltype(u_tag)::value, AF = (U + 1) & 1, RF = (U + 1) & 1;
    if (q < 16) {
      if constexpr (FIN) {
        const int j = q >> 2, k4 = q & 3, f = c - 2 + rr;
        if (k4 == 0) {
          float r4[4] = {0.f, 0.f, 0.f, 0.f};
          if constexpr (HAS_RES) {
            r4[0] = lo16<PREC>(rres[RF][j].x); r4[1] = hi16<PREC>(rres[RF][j].x);
            r4[2] = lo16<PREC>(rres[RF][j].y); r4[3] = hi16<PREC>(rres[RF][j].y);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            fv[k] = (acc[AF][4 * j + k] + addv[4 * j + k]) + r4[k];
            r6_pin(fv[k]);
          }
        }
        if (k4 == 1)
          *reinterpret_cast<uint2*>(p.dst + obase + j * plane + (size_t)f * p.w * 8) =
              make_uint2(pack2<PREC>(fv[0], fv[1]), pack2<PREC>(fv[2], fv[3]));
        if (k4 == 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            s1[4 * j + k] += fv[k];
            r6_pin(s1[4 * j + k]);
          }
        }
        if (k4 == 3) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            s2[4 * j + k] += fv[k] * fv[k];
            r6_pin(s2[4 * j + k]);
          }
        }
      }
    } else if (q < 58) {
      const int it = (q - 16) / 21, k21 = (q - 16) % 21;  // item (row c + 3 + it), piece within it
      const int i = c + 3 + it;
      if (k21 < 8) {
        if constexpr (ACT == 2) {
          const unsigned wv = word(xq[U][it], k21 >> 1);
          tv[it][k21] = ((k21 & 1) ? hi16<PREC>(wv) : lo16<PREC>(wv)) * scv[k21] + shv[k21];
          ev[it][k21] = __expf(-tv[it][k21]);
          r6_pin(tv[it][k21]);
          r6_pin(ev[it][k21]);
        }
      } else if (k21 < 16) {
        if constexpr (ACT == 2) {
          const int j = k21 - 8;
          ev[it][j] = tv[it][j] * __builtin_amdgcn_rcpf(1.0f + ev[it][j]);
          r6_pin(ev[it][j]);
        }
      } else if (k21 < 20) {
        const int jp = k21 - 16;
        const unsigned keep = ((unsigned)i < (unsigned)p.h) ? 0xFFFFFFFFu : 0u;
        if constexpr (ACT == 2) ow[it][jp] = pack2<PREC>(ev[it][2 * jp], ev[it][2 * jp + 1]) & keep;
        else ow[it][jp] = word(xq[U][it], jp) & keep;
        r6_pin(ow[it][jp]);
      } else {
        *reinterpret_cast<uint4*>(xl + slot_of(i) * R6_ROW_HALFS + m_lds) = make_uint4(ow[it][0], ow[it][1], ow[it][2], ow[it][3]);
      }
    } else if (q == 58) {
      if constexpr (ACT == 2) {
        th = lo16<PREC>((unsigned)hq[U]) * h_sc + h_sh;
        eh = __expf(-th);
        r6_pin(th);
        r6_pin(eh);
      }
    } else if (q == 59) {
      if constexpr (ACT == 2) {
        eh = th * __builtin_amdgcn_rcpf(1.0f + eh);
        r6_pin(eh);
      }
    } else if (q == 60) {
      const int i = c + 3 + h_row;
      const unsigned short km = (unsigned short)(h_keep & (((unsigned)i < (unsigned)p.h) ? 0xFFFF : 0));
      unsigned short hv = hq[U];
      if constexpr (ACT == 2) hv = (unsigned short)(pack2<PREC>(eh, 0.f) & 0xFFFFu);
      xl[slot_of(i) * R6_ROW_HALFS + h_lds] = (unsigned short)(hv & km);
    } else if (q == 61) {
      step_load(c + 9, (U + 3) & 3);
    } else if (q == 62) {
      res_load(c + rr, U & 1);
    }
  };
  auto step = [&](auto fin_tag, auto u_tag, int c) {
    constexpr int U = decltype(u_tag)::value, A = U & 1;
    const int srow = c + rr;  // this wave's output row
    r6_half8 bf[2];
    int so[3];                // ring slot offsets of input rows srow - 1, srow, srow + 1
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) so[dy] = slot_of(srow + dy - 1) * R6_ROW_HALFS;
    auto frag = [&](int s, int s2) {
      const int q = s / 9, tap = s % 9, dy = tap / 3, dx = tap % 3;
      bf[s2] = *reinterpret_cast<const r6_half8*>(bl + so[dy] + q * 2 * R6_COLS * 8 + dx * 8);
    };
    frag(0, 0);
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      const int fs = s & 1;
      if (s < 35) frag(s + 1, fs ^ 1);
      if (s == 0) r6_mma0<PREC>(acc[A], wa[0][0], bf[0]);
      else r6_mma<PREC>(acc[A], wa[s / 9][s % 9], bf[fs]);
#pragma unroll
      for (int q = 0; q < 63; ++q)
        if ((q + 16) * 36 / 80 == s + 7 || (s == 35 && (q + 16) * 36 / 80 > 42)) piece(fin_tag, u_tag, q, c);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  // flush of the 8-row x 32-column statistics tile of rows t .. t + 7: the two row-waves of a cout tile add their halves through LDS
  auto flush = [&](int t) {
    if (rr == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        stl[(mt * 32 + r) * 64 + lane] = s1[r];
        stl[(mt * 32 + 16 + r) * 64 + lane] = s2[r];
      }
    }
    __syncthreads();
    if (rr == 0 && p.stats != nullptr) {
      const int tile = (t >> 3) * tiles_x + cx;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a = half_wave_sum(s1[r] + stl[(mt * 32 + r) * 64 + lane]);
        const float b = half_wave_sum(s2[r] + stl[(mt * 32 + 16 + r) * 64 + lane]);
        if (l31 == 31) {
          double* o = p.stats + (((size_t)n * 64 + mt * 32 + 8 * (r >> 2) + 4 * half + (r & 3)) * ntile + tile) * 2;
          o[0] = (double)a;
          o[1] = (double)b;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = s2[r] = 0.f;
    __syncthreads();
  };

  // ---- prologue: input rows r0 - 1 .. r0 + 2 into the ring; register sets 0 .. 2 for the first three steps --------------
  step_load(r0 - 1, 0);
  commit_now(r0 - 1, 0);
  step_load(r0 + 1, 1);
  commit_now(r0 + 1, 1);
  step_load(r0 + 3, 0);
  step_load(r0 + 5, 1);
  step_load(r0 + 7, 2);
  res_load(r0 + rr, 0);   // (step 0 reloads it: harmless)
  __syncthreads();

  // ---- the band: four steps (eight rows) per trip; a tile's statistics are complete one step into the next trip ---------
  step(std::false_type{}, std::integral_constant<int, 0>{}, r0);
  for (int t = r0; t < r1; t += 8) {
    if (t > r0) {
      step(std::true_type{}, std::integral_constant<int, 0>{}, t);
      flush(t - 8);
    }
    step(std::true_type{}, std::integral_constant<int, 1>{}, t + 2);
    step(std::true_type{}, std::integral_constant<int, 2>{}, t + 4);
    step(std::true_type{}, std::integral_constant<int, 3>{}, t + 6);
  }
  step(std::true_type{}, std::integral_constant<int, 0>{}, r1);  // (two rows past the band: finishes the band's last two rows)
  flush(r1 - 8);
}

// shape rule only (never the batch): a layer is served by this kernel at every batch size or at none
bool conv_rs16_eligible(const dsg_conv_args* a, int hout, int wout) {
  if (!g_h2.enabled || !g_h2.rs16 || (a->compute_dtype != DSG_BF16 && a->compute_dtype != DSG_F16)) return false;
#ifdef R6_DEV_BF16_ONLY
  if (a->compute_dtype != DSG_BF16) return false;
#endif
  if (a->ksize != 3 || a->stride != 1 || a->upsample || a->pool2 || a->src_layout != 1 || a->dst_layout != 1) return false;
  if (a->c0 != 64 || a->c1 != 0 || a->cout != 64 || a->weight_h2 == nullptr) return false;
  if (!((a->gn_scale_shift && a->silu) || (!a->gn_scale_shift && !a->silu))) return false;
  if (a->weight_h2_cout_stride && a->weight_h2_cout_stride != 64) return false;
  if (a->sc_weight_h2 != nullptr || a->src_operand != nullptr) return false;
  if (wout % 32 != 0 || hout % 8 != 0 || hout < 64 || wout < 64) return false;
  return true;
}

template <int PREC>
static int conv_rs16_launch_t(const ConvRs16P& p, bool act, int grid, hipStream_t st) {
  static bool raised = false;
  if (!raised) {
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs16_kernel<PREC, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs16_kernel<PREC, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs16_kernel<PREC, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs16_kernel<PREC, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised = true;
  }
  const dim3 g(grid), b(256);
  const size_t lds = (size_t)R6_LDS_BYTES;
  if (act && p.res) hipLaunchKernelGGL((conv_rs16_kernel<PREC, 2, true>), g, b, lds, st, p);
  else if (act) hipLaunchKernelGGL((conv_rs16_kernel<PREC, 2, false>), g, b, lds, st, p);
  else if (p.res) hipLaunchKernelGGL((conv_rs16_kernel<PREC, 0, true>), g, b, lds, st, p);
  else hipLaunchKernelGGL((conv_rs16_kernel<PREC, 0, false>), g, b, lds, st, p);
  return DSG_OK;
}

int conv_rs16_launch(const dsg_conv_args* a, int hout, int wout, hipStream_t st) {
  ConvRs16P p;
  p.src = reinterpret_cast<const unsigned short*>(a->src0);
  p.wh = static_cast<const unsigned short*>(a->weight_h2);
  p.ss = a->gn_scale_shift;
  p.bias = a->bias;
  p.temb = a->temb;
  p.temb_stride = a->temb_stride;
  p.res = reinterpret_cast<const unsigned short*>(a->residual);
  p.dst = reinterpret_cast<unsigned short*>(a->dst);
  p.stats = a->stats_out;
  p.n = a->n; p.h = hout; p.w = wout;
  p.band = hout % 32 == 0 ? 32 : (hout % 16 == 0 ? 16 : 8);
  int pi = -1;
  if (prof_on()) {
    const double px = (double)p.n * hout * wout;
    pi = prof_begin(36, 2.0 * px * 64 * 64 * 9, 2.0 * px * 64 * (p.res ? 3.0 : 2.0) + 2.0 * 64 * 64 * 9, st);
  }
  const int grid = (wout / 32) * (hout / p.band) * p.n;
#ifdef R6_DEV_BF16_ONLY
  const int rc = conv_rs16_launch_t<1>(p, a->gn_scale_shift != nullptr, grid, st);
#else
  const int rc = a->compute_dtype == DSG_BF16 ? conv_rs16_launch_t<1>(p, a->gn_scale_shift != nullptr, grid, st)
                                              : conv_rs16_launch_t<2>(p, a->gn_scale_shift != nullptr, grid, st);
#endif
  if (rc != DSG_OK) return rc;
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

void conv_h2_set_rs16(int v) { g_h2.rs16 = v; ++g_h2.epoch; }

}  // namespace dsg
