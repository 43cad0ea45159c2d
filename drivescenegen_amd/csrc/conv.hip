// Fused implicit-GEMM convolution for gfx950 (MI355X), fp32 on the f32-input matrix cores.
//
// Replaces every nn.Conv2d / attention nn.Linear call of diffusers' UNet2DModel.forward as used by
// DriveSceneGen (reference: DriveSceneGen/scripts/train.py:39-57 builds the net,
// DriveSceneGen/pipeline/training_pipeline.py:84 runs it; semantics SURVEY.md App. A.2), and -- with
// transposed/flipped weights -- the data-gradient convolutions of the backward pass
// (training_pipeline.py:86 `accelerator.backward(loss)`).
//
// GEMM view:  D[cout][pixel] = sum_{cin,tap} W[cin][tap][cout] * X[cin][pixel + tap]
//   A operand = weights  (M = cout)    -> v_mfma_f32_32x32x2_f32, exact f32 (fmaf chain in k order)
//   B operand = pixels   (N = 32 consecutive pixels of one output row -> 128-B coalesced stores)
// A workgroup (4 waves) owns BM = 32*MT couts x (8 rows x 32 cols) output pixels of one image.
// Per K-chunk of KC input channels it stages, through registers, the weight slab [KC][taps][BM] and
// the input halo patch [KC][PH][PW] into LDS -- GroupNorm-apply + SiLU of the previous norm is folded
// into that staging pass; the [x || skip] concat, the nearest-x2 upsample (gather mode 1) and the
// zero-stuffed x2 upsample of a stride-2 conv's data gradient (gather mode 2) into its gather -- then all
// 9 taps are served from the single patch (no im2col blow-up).
//
// Pipeline: double-buffered LDS, ONE barrier per K-chunk.  While the wave issues the MFMAs of chunk q
// from buffer q&1, the staging of chunk q+1 (ds_write into the other buffer) and the global loads of
// chunk q+2 are sliced into NP pieces and interleaved between the MFMA groups in program order (the f32
// MFMA leaves 15 of 16 issue slots free).  The chunk body is branch-free (clamped loads, selects, stores
// into padded LDS slabs) so it stays ONE basic block and the scheduler runs ds_reads ahead of the MFMAs.
// Epilogue: + bias (+ time-embedding column) (+ residual), 128-B row stores; optional 2x2 sum-pool
// (the adjoint of the nearest-x2 upsample).
#include "dsg_h16.h"
#include <cstdlib>
#include <algorithm>

namespace dsg {

bool prof_on();
int prof_begin(int kid, double flops, double bytes, hipStream_t st);
void prof_end(int idx, hipStream_t st);

struct ConvP {
  const float* src0;
  const float* src1;
  int c0, c1, cin;
  int n, hin, win;  // source dims
  int hc, wc;       // conv-input dims (after the optional x2 gather mode)
  int hout, wout;   // conv output dims (before the optional pool)
  int cout;         // real output channels
  int wstride;      // row stride of the weight matrix (>= cout, multiple of 32 for the MFMA path)
  const float* w;
  const float* bias;
  const float* ss;
  int silu;
  const float* temb;
  int temb_stride;
  const float* res;
  float* dst;
  int pool;  // 1: 2x2 sum-pool in the epilogue; dst is [N, cout, hout/2, wout/2]
  int tiles_x, tiles_y;
  int sblk, dblk;  // 1: sources / (dst, residual) are channel-blocked [N][C/8][H][W][8] (dsg_conv_args.*_layout)
  int dt;          // dsg_dtype of the channel-blocked tensors: 0 fp32, 1 bf16, 2 fp16 ([N,C,H,W] tensors are always fp32)
};

constexpr int TH = 8;   // output rows per workgroup
constexpr int TW = 32;  // output cols per workgroup (= MFMA N)

template <int KS, int STRIDE, int KC>
struct ConvGeom {
  static constexpr int TAPS = KS * KS;
  static constexpr int PH = (TH - 1) * STRIDE + KS;
  static constexpr int PW = (TW - 1) * STRIDE + KS;
  static constexpr int PSZ = PH * PW;
  static constexpr int XN = KC * PSZ;
};

__device__ __forceinline__ float silu_fast(float x) {
  // x * 1/(1+exp(-x)) with v_exp_f32 / v_rcp_f32 (a few ulp; far inside the 1e-4 forward tolerance)
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// GM: gather mode 0 plain, 1 nearest x2 (source pixel [y>>1][x>>1]), 2 zero-stuffed x2 (source pixel
// [y/2][x/2] at even (y, x), zero elsewhere)
// IO16: bit 0: the (channel-blocked) sources are 16-bit values of type p.dt; bit 1: dst / residual are.  The arithmetic
// stays the exact fp32 MFMA chain: this is how conv_in (fp32 [N,C,H,W] image -> 16-bit blocked activations) and the
// shapes the split kernels do not take run in the mixed-precision modes.
template <int KS, int STRIDE, int GM, int MT, int KC, int IO16 = 0>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvP p) {
  constexpr bool S16 = (IO16 & 1) != 0, D16 = (IO16 & 2) != 0;
  using G = ConvGeom<KS, STRIDE, KC>;
  constexpr int TAPS = G::TAPS, PW = G::PW, PSZ = G::PSZ, XN = G::XN;
  constexpr int BM = MT * 32;
  constexpr int TPC = 256 / KC;               // threads staging one channel of the patch
  constexpr int NE = (PSZ + TPC - 1) / TPC;   // patch elements per thread per chunk (all of ONE channel)
  constexpr int WN4 = KC * TAPS * BM / 4;
  constexpr int NW = (WN4 + 255) / 256;
  constexpr int WSZ = NW * 256 * 4;           // padded weight slab (floats)
  constexpr int XSZ = (XN + 4 + 3) & ~3;      // patch + one dump slot for masked-off lanes
  constexpr int BUF = WSZ + XSZ;
  constexpr int NIT = (KC / 2) * TAPS;        // MFMA groups per chunk
  constexpr int NP = NE + NW;                 // staging pieces per chunk

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int m0 = blockIdx.y * BM;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * STRIDE - KS / 2, ix0 = ox0 * STRIDE - KS / 2;
  const int plane = p.hin * p.win;
  const int nq = (p.cin + KC - 1) / KC;

  // This thread stages channel `tc` of every chunk: NE positions of its halo patch.
  const int tc = tid / TPC;
  const int tr = tid - tc * TPC;
  int goff[NE];        // clamped gather offsets within the channel plane
  int loff[NE];        // LDS offset in the patch slab (dump slot when the position is past the patch)
  unsigned valid = 0;  // bit i: position i is inside the image (else zero padding)
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int r = tr + TPC * i;
    int off = 0;
    if (r < PSZ) {
      const int py = r / PW;
      const int px = r - py * PW;
      const int gy = iy0 + py, gx = ix0 + px;
      bool ok = gy >= 0 && gy < p.hc && gx >= 0 && gx < p.wc;
      if (GM == 2) ok = ok && ((gy | gx) & 1) == 0;
      if (ok) {
        const int sy = GM ? (gy >> 1) : gy;
        const int sx = GM ? (gx >> 1) : gx;
        off = (sy * p.win + sx) * (p.sblk ? 8 : 1);  // blocked: a pixel's 8 channels are adjacent
        valid |= 1u << i;
      }
    }
    goff[i] = off;
    loff[i] = r < PSZ ? tc * PSZ + r : XN;
  }
  const bool has_ss = p.ss != nullptr;
  const bool do_silu = has_ss && p.silu;
  const float* ssg = has_ss ? p.ss + (size_t)n * p.cin * 2 : nullptr;

  float xr[NE];
  float wr[NW][4];
  float sc_r = 1.f, sh_r = 0.f;  // GroupNorm scale/shift of this thread's channel in the chunk held in xr
  bool cok_r = true;             // ... and whether that channel exists (cin not a multiple of KC)

  // chunk q covers channels [q*KC, q*KC+KC) of ONE source (c0 % KC == 0 when there are two)
  auto chan_of = [&](int q, bool& ok) -> const float* {
    int c = q * KC + tc;
    ok = c < p.cin;
    c = min(c, p.cin - 1);
    const float* sp = p.src0;
    int cs = p.c0;
    if (c >= p.c0) {
      sp = p.src1;
      cs = p.c1;
      c -= p.c0;
    }
    // blocked: channel c lives at offset c % 8 inside the block that starts where plane (c & ~7) would
    // (16-bit sources: the returned pointer is used as an unsigned short*, so the element offset is the same)
    const size_t eo = p.sblk ? ((size_t)n * cs + (c & ~7)) * plane + (c & 7) : ((size_t)n * cs + c) * plane;
    return S16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(sp) + eo) : sp + eo;
  };
  auto w_of = [&](int q) -> const float* { return p.w + (size_t)q * (KC * TAPS) * p.wstride + m0; };
  const int wrow_max = p.cin * TAPS - 1;
  // piece pc < NE: one patch element per thread; pc >= NE: one float4 of the weight slab
  auto load_piece = [&](int pc, int q, const float* sp, const float* wp) {
    if (pc < NE) {
      if constexpr (S16) xr[pc] = ld16(reinterpret_cast<const unsigned short*>(sp) + goff[pc], p.dt);
      else xr[pc] = sp[goff[pc]];
    } else {
      const int i = pc - NE;
      const int idx = min(tid + 256 * i, WN4 - 1);
      const int row = idx / (BM / 4);
      const int c4 = idx - row * (BM / 4);
      const int grow = min(q * (KC * TAPS) + row, wrow_max) - q * (KC * TAPS);  // rows past cin: finite junk x 0
      const float4 t4 = *reinterpret_cast<const float4*>(wp + (ptrdiff_t)grow * p.wstride + c4 * 4);
      wr[i][0] = t4.x; wr[i][1] = t4.y; wr[i][2] = t4.z; wr[i][3] = t4.w;
    }
  };
  auto load_ss = [&](int q) {
    if (has_ss) {
      const int c = min(q * KC + tc, p.cin - 1);
      const float2 s2 = *reinterpret_cast<const float2*>(ssg + 2 * c);
      sc_r = s2.x;
      sh_r = s2.y;
    }
  };
  auto commit_piece = [&](int pc, float* Wd, float* Xd) {
    if (pc < NE) {
      float v = xr[pc] * sc_r + sh_r;
      const float sv = silu_fast(v);
      v = do_silu ? sv : v;
      v = (((valid >> pc) & 1u) && cok_r) ? v : 0.f;
      Xd[loff[pc]] = v;
    } else {
      const int i = pc - NE;
      reinterpret_cast<float4*>(Wd)[tid + 256 * i] = make_float4(wr[i][0], wr[i][1], wr[i][2], wr[i][3]);
    }
  };

  f32x16 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // prologue: chunk 0 into buffer 0, chunk 1 into registers
  {
    const float* sp = chan_of(0, cok_r);
    const float* wp = w_of(0);
    load_ss(0);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc, 0, sp, wp);
  }
#pragma unroll
  for (int pc = 0; pc < NP; ++pc) commit_piece(pc, smem, smem + WSZ);
  {
    const int q1 = min(1, nq - 1);
    const float* sp = chan_of(q1, cok_r);
    const float* wp = w_of(q1);
    load_ss(q1);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc, q1, sp, wp);
  }
  __syncthreads();

  const int wl_off = half * (TAPS * BM) + l31;
  const int xl_off = WSZ + half * PSZ + (wave * 2 * STRIDE) * PW + l31 * STRIDE;

  for (int q = 0; q < nq; ++q) {
    const float* cur = smem + (q & 1) * BUF;
    float* nxt = smem + ((q & 1) ^ 1) * BUF;
    const float* wl = cur + wl_off;
    const float* xl = cur + xl_off;
    // registers hold chunk min(q+1, nq-1); it is committed to the other buffer while chunk min(q+2, nq-1)
    // is loaded behind it (redundant, harmless work on the last two iterations)
    const int ql = min(q + 2, nq - 1);
    bool cok_n;
    const float* spn = chan_of(ql, cok_n);
    const float* wpn = w_of(ql);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) {
        if (pc * NIT / NP == it) {
          commit_piece(pc, nxt, nxt + WSZ);
          load_piece(pc, ql, spn, wpn);
          if (pc == NE - 1) {  // last patch piece of the committed chunk: switch to the next chunk's scalars
            load_ss(ql);
            cok_r = cok_n;
          }
        }
      }
      const int cp = it / TAPS, tap = it % TAPS;
      const int dy = tap / KS, dx = tap % KS;
      float a[MT], b[2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = wl[(2 * cp) * (TAPS * BM) + tap * BM + mt * 32];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) b[nt] = xl[(2 * cp) * PSZ + (nt * STRIDE + dy) * PW + dx];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
    __syncthreads();
  }

  // Epilogue. C/D layout of the 32x32 tile: col (pixel) = lane&31, row (cout) = (r&3) + 8*(r>>2) + 4*half.
  const int x = ox0 + l31;
  const bool has_t = p.temb != nullptr;
  auto oidx = [&](int co, int y, int xx) -> size_t {
    return p.dblk ? ((size_t)n * p.cout + (co & ~7)) * p.hout * p.wout + ((size_t)y * p.wout + xx) * 8 + (co & 7)
                  : (((size_t)n * p.cout + co) * p.hout + y) * p.wout + xx;
  };
  if (!p.pool) {
    // all residual loads first (in flight together), then add + store
    const bool has_r = p.res != nullptr;
    float rv[MT][16][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = min(m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.cout - 1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int y = oy0 + wave * 2 + nt;
          if constexpr (D16)
            rv[mt][r][nt] = has_r ? ld16(reinterpret_cast<const unsigned short*>(p.res) + oidx(co, y, min(x, p.wout - 1)), p.dt) : 0.f;
          else
            rv[mt][r][nt] = has_r ? p.res[oidx(co, y, min(x, p.wout - 1))] : 0.f;
        }
      }
    if (p.dblk) {
      // channel-blocked dst: a register group (r >> 2) is four consecutive channels of one pixel = one 16-byte store
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int co0 = m0 + mt * 32 + 8 * rg + 4 * half;
          if (co0 < p.cout && x < p.wout) {  // (cout % 8 == 0: the group is valid as a whole)
            float add[4], tv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              add[j] = p.bias ? p.bias[co0 + j] : 0.f;
              tv[j] = has_t ? p.temb[(size_t)n * p.temb_stride + co0 + j] : 0.f;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const int y = oy0 + wave * 2 + nt;
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {  // (the same order of additions as the [N,C,H,W] branch)
                v[j] = acc[mt][nt][4 * rg + j] + add[j];
                if (has_t) v[j] = v[j] + tv[j];
                if (has_r) v[j] = v[j] + rv[mt][4 * rg + j][nt];
              }
              if constexpr (D16)
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.dst) + oidx(co0, y, x)) =
                    make_uint2(word_pack(v[0], v[1], p.dt), word_pack(v[2], v[3], p.dt));
              else
                *reinterpret_cast<float4*>(p.dst + oidx(co0, y, x)) = make_float4(v[0], v[1], v[2], v[3]);
            }
          }
        }
    } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < p.cout && x < p.wout) {  // (x >= wout only on maps narrower than a tile)
          const float add = p.bias ? p.bias[co] : 0.f;
          const float tv = has_t ? p.temb[(size_t)n * p.temb_stride + co] : 0.f;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int y = oy0 + wave * 2 + nt;
            const size_t idx = oidx(co, y, x);
            float v = acc[mt][nt][r] + add;
            if (has_t) v = v + tv;
            if (has_r) v = v + rv[mt][r][nt];
            p.dst[idx] = v;
          }
        }
      }
    }
    }
  } else {
    // 2x2 sum-pool: the wave's two rows are one pooled row; column pairs are adjacent lanes
    const int ho2 = p.hout >> 1, wo2 = p.wout >> 1;
    const int y2 = (oy0 >> 1) + wave;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[mt][0][r] + acc[mt][1][r];
        v += __shfl_xor(v, 1, 64);
        if (co < p.cout && (l31 & 1) == 0) {
          const size_t idx = (((size_t)n * p.cout + co) * ho2 + y2) * wo2 + (x >> 1);
          float o = v + (p.bias ? 4.f * p.bias[co] : 0.f);
          if (p.res) o = o + p.res[idx];
          p.dst[idx] = o;
        }
      }
    }
  }
}

// General VALU fallback: any channel counts / sizes.  One thread per output pixel, COB couts per thread.
template <int COB>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvP p, int ks, int stride, int gm) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.z;
  const int co0 = blockIdx.y * COB;
  if (pix >= p.hout * p.wout) return;
  const int oy = pix / p.wout, ox = pix - oy * p.wout;
  const int pad = ks / 2;
  const int plane = p.hin * p.win;
  float acc[COB];
#pragma unroll
  for (int j = 0; j < COB; ++j) acc[j] = 0.f;
  const int taps = ks * ks;
  for (int c = 0; c < p.cin; ++c) {
    const float* sp = (c < p.c0) ? p.src0 + ((size_t)n * p.c0 + c) * plane
                                 : p.src1 + ((size_t)n * p.c1 + (c - p.c0)) * plane;
    float sc = 1.f, sh = 0.f;
    if (p.ss) {
      sc = p.ss[((size_t)n * p.cin + c) * 2];
      sh = p.ss[((size_t)n * p.cin + c) * 2 + 1];
    }
    for (int dy = 0; dy < ks; ++dy) {
      const int gy = oy * stride - pad + dy;
      if (gy < 0 || gy >= p.hc) continue;
      for (int dx = 0; dx < ks; ++dx) {
        const int gx = ox * stride - pad + dx;
        if (gx < 0 || gx >= p.wc) continue;
        if (gm == 2 && ((gy | gx) & 1)) continue;
        const int sy = gm ? (gy >> 1) : gy, sx = gm ? (gx >> 1) : gx;
        float v = sp[sy * p.win + sx];
        if (p.ss) {
          v = v * sc + sh;
          if (p.silu) v = silu_f(v);
        }
        const float* wrow = p.w + ((size_t)c * taps + dy * ks + dx) * p.wstride + co0;
#pragma unroll
        for (int j = 0; j < COB; ++j)
          if (co0 + j < p.cout) acc[j] = fmaf(wrow[j], v, acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < COB; ++j) {
    const int co = co0 + j;
    if (co >= p.cout) break;
    const size_t idx = ((size_t)n * p.cout + co) * p.hout * p.wout + pix;
    float v = acc[j] + (p.bias ? p.bias[co] : 0.f);
    if (p.temb) v = v + p.temb[(size_t)n * p.temb_stride + co];
    if (p.res) v = v + p.res[idx];
    p.dst[idx] = v;
  }
}

// mode 0: forward layout  dst[(ci*taps + tap)*cout_total + cout_off + co] = w[co][ci][tap]
// mode 1: data-gradient layout (transposed, taps flipped)  dst[(co*taps + (taps-1-tap))*cout_total + cout_off + ci]
__global__ void weight_relayout_kernel(const float* __restrict__ w, float* __restrict__ dst, int cout, int cin,
                                       int taps, int cout_total, int cout_off, int mode) {
  const int64_t total = (int64_t)cout * cin * taps;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (mode == 0) {
      const int co = (int)(i % cout);
      const int64_t r = i / cout;
      const int tap = (int)(r % taps);
      const int ci = (int)(r / taps);
      dst[((int64_t)ci * taps + tap) * cout_total + cout_off + co] = w[((int64_t)co * cin + ci) * taps + tap];
    } else {
      const int ci = (int)(i % cin);
      const int64_t r = i / cin;
      const int tap = (int)(r % taps);
      const int co = (int)(r / taps);
      dst[((int64_t)co * taps + (taps - 1 - tap)) * cout_total + cout_off + ci] =
          w[((int64_t)co * cin + ci) * taps + tap];
    }
  }
}

bool conv_h2_eligible(const dsg_conv_args* a, int hout, int wout);
bool conv_h2_s2(const dsg_conv_args* a, int hout, int wout);
int conv_h2_launch(const dsg_conv_args* a, int hout, int wout, hipStream_t st);
int conv_h2_stats_tiles(const dsg_conv_args* a, int hout, int wout);
// conv_in.hip: fp32 [N,C<=8,H,W] image -> channel-blocked activations, every compute_dtype
bool conv_in_eligible(const dsg_conv_args* a, int hout, int wout);
int conv_in_stats_tiles(const dsg_conv_args* a, int hout, int wout);
int conv_in_launch(const dsg_conv_args* a, int hout, int wout, hipStream_t st);
void conv_in_set_enabled(int v);
// conv_out.hip: normalised + activated channel-blocked activations -> fp32 [N,C<=8,H,W] image, every compute_dtype
bool conv_out_eligible(const dsg_conv_args* a, int hout, int wout);
int conv_out_launch(const dsg_conv_args* a, int hout, int wout, hipStream_t st);
void conv_out_set_enabled(int v);
void conv_h2_set_enabled(int on);
void conv_h2_set_rows(int r);
void conv_h2_set_stats(int on);
void conv_h2_set_waves(int w);
void conv_h2_set_pw_occ2(int v);
void conv_h2_set_s2(int v);
void conv_h2_set_bm32(int v);
void conv_h2_set_bm32_small(int v);
void conv_h2_set_bm128(int v);
void conv_h2_set_splitk(int v);
void conv_h2_set_ws2(int v);
void conv_h2_set_fuse_sc(int v);
int conv_h2_get_fuse_sc();
void conv_h2_set_pre(int v);
void conv_h2_set_narrow(int v);
void conv_h2_set_splitk_mid(int v);
void conv_h2_set_rows_rule(int v);
void conv_h2_set_gnb(int v);
void conv_h2_set_s2_nchw(int v);
void conv_h2_set_gnb_bm64(int v);
void attention_set_bwd_split(int v);
bool conv_h2_gnb_ok(const dsg_conv_args* a, int hout, int wout);
void conv_h2_set_pre_min_ct(int v);
bool conv_h2_takes_operand(const dsg_conv_args* a, int hout, int wout, bool wanted);
void attention_set_blocked(int v);
bool conv_h2_sc_fusable(const dsg_conv_args* a, int hout, int wout);
int conv_h2_splitk_slices(const dsg_conv_args* a, int hout, int wout, int* stat_splits);
void unet_set_blocked(int v);
void attention_set_mfma(int v);
void wgrad_h2_set_enabled(int on);
void conv_wgrad16_set_wide(int v);
void conv_wgrad16_set_fold(int v);
void wgrad_h2_set_wide(int v);
void conv_wgrad16_set_pw(int v);
void conv_h2_set_fold(int on);

static int g_conv_fewout = 1;  // VALU kernel for cout <= 4 (tuning key 10: A/B against the zero-padded MFMA tile)
static int g_conv_kc = 0;  // K-chunk of the 3x3 stride-1 kernel: 4 | 8 | 0 = by grid size (measured, r01)

template <int KS, int STRIDE, int GM, int MT, int KC, int IO16 = 0>
static int launch_mfma(const ConvP& p, hipStream_t st) {
  using G = ConvGeom<KS, STRIDE, KC>;
  constexpr int NW = (KC * G::TAPS * MT * 32 / 4 + 255) / 256;
  constexpr int XSZ = (G::XN + 4 + 3) & ~3;
  const size_t lds = (size_t)(2 * (NW * 1024 + XSZ)) * sizeof(float);
  dim3 grid(p.tiles_x * p.tiles_y * p.n, (p.cout + MT * 32 - 1) / (MT * 32));
  auto kern = conv_mfma_kernel<KS, STRIDE, GM, MT, KC, IO16>;
  static bool raised = false;
  if (!raised) {
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    raised = true;
  }
  int pi = -1;
  if (prof_on()) {
    const double px = (double)p.n * p.hout * p.wout;
    const double flops = 2.0 * px * p.cout * p.cin * G::TAPS;
    // algorithmic bytes: input read once, weights once, output written once (+ residual read)
    const double bytes = 4.0 * ((double)p.n * p.cin * p.hin * p.win + (double)p.cin * G::TAPS * p.cout +
                                px * p.cout * (p.res ? 2.0 : 1.0));
    pi = prof_begin(KS == 1 ? 3 : (STRIDE == 2 ? 2 : (GM ? 1 : 0)), flops, bytes, st);
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}


// conv_out (64 -> 3/4 channels, GroupNorm + SiLU in front): with so few output channels the matrix cores would
// compute 8x more columns than exist, so this one runs on the VALU from an LDS-staged patch: a workgroup owns a
// 16-row x 32-column tile, stages 16 channels of the activated 18 x 34 halo patch per step and every thread holds
// 2 pixels x 4 output channels (each patch value read from LDS feeds 4 FMAs, each weight quad 2 pixels).
// Weights are the engine layout [cin][9][wstride]; fp32 fmaf chains in channel-major order.
constexpr int FO_TH = 16, FO_PW = 35, FO_PH = 18, FO_KC = 16, FO_CST = FO_PH * FO_PW;  // (35: odd row stride)

__global__ __launch_bounds__(256, 2) void conv_fewout_kernel(ConvP p) {
  __shared__ float xs[FO_KC * FO_CST];
  __shared__ __attribute__((aligned(16))) float wsm[FO_KC * 9 * 4];
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int oy0 = ty * FO_TH, ox0 = tx * TW;
  const int plane = p.hin * p.win;
  const int col = tid & 31, r0 = tid >> 5;  // pixels (r0, col) and (r0 + 8, col)
  float acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const bool has_ss = p.ss != nullptr;
  // this thread's (up to 3) patch positions: global offset, LDS offset, inside-the-image flag -- once per tile
  int goff[3], loff[3];
  bool ok[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int pos = tid + 256 * k;
    const int py = pos / 34, px = pos - py * 34;
    const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
    ok[k] = pos < FO_PH * 34 && gy >= 0 && gy < p.hc && gx >= 0 && gx < p.wc;
    goff[k] = ok[k] ? (gy * p.win + gx) * (p.sblk ? 8 : 1) : 0;
    loff[k] = pos < FO_PH * 34 ? py * FO_PW + px : -1;
  }
  // the raw values of the next 16 channels are fetched into registers while the current 16 are being used
  float xr[FO_KC][3];
  auto fetch = [&](int c0) {
    const float* spb = (c0 < p.c0) ? p.src0 + ((size_t)n * p.c0 + c0) * plane
                                   : p.src1 + ((size_t)n * p.c1 + (c0 - p.c0)) * plane;  // (uniform: c0 % 16 == 0)
    if (p.sblk && p.dt) {
      // 16-bit blocked sources (mixed-precision modes): a position's 8 channels are one 16-byte load
      const unsigned short* sph = reinterpret_cast<const unsigned short*>((c0 < p.c0) ? p.src0 : p.src1) +
                                  ((c0 < p.c0) ? ((size_t)n * p.c0 + c0) : ((size_t)n * p.c1 + (c0 - p.c0))) * plane;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const uint4 q = *reinterpret_cast<const uint4*>(sph + (size_t)b * 8 * plane + goff[k]);
          xr[8 * b + 0][k] = word_lo(q.x, p.dt); xr[8 * b + 1][k] = word_hi(q.x, p.dt);
          xr[8 * b + 2][k] = word_lo(q.y, p.dt); xr[8 * b + 3][k] = word_hi(q.y, p.dt);
          xr[8 * b + 4][k] = word_lo(q.z, p.dt); xr[8 * b + 5][k] = word_hi(q.z, p.dt);
          xr[8 * b + 6][k] = word_lo(q.w, p.dt); xr[8 * b + 7][k] = word_hi(q.w, p.dt);
        }
    } else if (p.sblk) {
      // blocked sources: the 16-channel chunk starts at the same address and is two channel blocks; a position's 8
      // channels are two 16-byte loads
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float4* q = reinterpret_cast<const float4*>(spb + (size_t)b * 8 * plane + goff[k]);
          const float4 lo = q[0], hi = q[1];
          xr[8 * b + 0][k] = lo.x; xr[8 * b + 1][k] = lo.y; xr[8 * b + 2][k] = lo.z; xr[8 * b + 3][k] = lo.w;
          xr[8 * b + 4][k] = hi.x; xr[8 * b + 5][k] = hi.y; xr[8 * b + 6][k] = hi.z; xr[8 * b + 7][k] = hi.w;
        }
    } else {
#pragma unroll
      for (int c = 0; c < FO_KC; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) xr[c][k] = spb[(size_t)c * plane + goff[k]];
    }
  };
  fetch(0);
  for (int c0 = 0; c0 < p.cin; c0 += FO_KC) {
    __syncthreads();
    // stage: 16 channels x 18 x 34 activated values (zero outside the image), and their 16 x 9 x 4 weights
    const float* ssb = has_ss ? p.ss + ((size_t)n * p.cin + c0) * 2 : nullptr;
#pragma unroll
    for (int c = 0; c < FO_KC; ++c) {
      float sc = 1.f, sh = 0.f;
      if (has_ss) {
        sc = ssb[2 * c];
        sh = ssb[2 * c + 1];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (loff[k] >= 0) {
          float v = xr[c][k];
          if (has_ss) {
            v = v * sc + sh;
            if (p.silu) v = silu_fast(v);
          }
          xs[c * FO_CST + loff[k]] = ok[k] ? v : 0.f;
        }
      }
    }
    for (int e = tid; e < FO_KC * 9 * 4; e += 256) {
      const int j = e & 3, ct = e >> 2;  // ct = c * 9 + tap
      wsm[e] = (j < p.cout) ? p.w[((size_t)c0 * 9 + ct) * p.wstride + j] : 0.f;
    }
    __syncthreads();
    if (c0 + FO_KC < p.cin) fetch(c0 + FO_KC);
#pragma unroll 2
    for (int c = 0; c < FO_KC; ++c) {
      const float* xp = xs + c * FO_CST + r0 * FO_PW + col;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const float4 wq = *reinterpret_cast<const float4*>(wsm + (c * 9 + t) * 4);
        const f2 w01 = {wq.x, wq.y}, w23 = {wq.z, wq.w};
        // VOLATILE, LDS address space, on purpose (round 4).  Left to itself hipcc 7.2 pairs these reads into ds_read2_b32 and then
        // feeds the packed FMAs below with `op_sel:[0,1,0]` (both lanes take the pair's HIGH dword).  On gfx950 a packed fp32 VALU op
        // of that form returns wrong values on lanes 48-63 while ANOTHER WAVE on the same SIMD issues v_mfma_f32_32x32x16_f16 (here: a wave
        // of another process; a kernel's own waves do it too, probe mode 18)
        // (tools/probes/probe_lds_read2.hip modes 13-15 next to probe_neighbour.hip mode 0; profiles/r04_race_under_load.txt): this
        // kernel was off by 1e-3 .. 1e-1 in ~90 % of its launches next to a second process of this library and bit-stable alone.
        // Single reads land in registers of their own and the FMAs take `op_sel_hi:[1,0,1]` (low dword twice), which is not
        // affected; tests/test_isa_policy.py fails any build of the library that contains the other form again.
        const volatile __attribute__((address_space(3))) float* xv = (const volatile __attribute__((address_space(3))) float*)xp;
        const float a0 = xv[(t / 3) * FO_PW + (t % 3)], a1 = xv[(t / 3 + 8) * FO_PW + (t % 3)];
        const f2 b0 = {a0, a0}, b1 = {a1, a1};
        f2& c00 = *reinterpret_cast<f2*>(&acc[0][0]);
        f2& c02 = *reinterpret_cast<f2*>(&acc[0][2]);
        f2& c10 = *reinterpret_cast<f2*>(&acc[1][0]);
        f2& c12 = *reinterpret_cast<f2*>(&acc[1][2]);
        c00 = __builtin_elementwise_fma(w01, b0, c00);  // (v_pk_fma_f32: two output channels per instruction)
        c02 = __builtin_elementwise_fma(w23, b0, c02);
        c10 = __builtin_elementwise_fma(w01, b1, c10);
        c12 = __builtin_elementwise_fma(w23, b1, c12);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int y = oy0 + r0 + 8 * i, x = ox0 + col;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < p.cout) {
        const size_t idx = (((size_t)n * p.cout + j) * p.hout + y) * p.wout + x;
        float v = acc[i][j] + (p.bias ? p.bias[j] : 0.f);
        if (p.temb) v = v + p.temb[(size_t)n * p.temb_stride + j];
        if (p.res) v = v + p.res[idx];
        p.dst[idx] = v;
      }
    }
  }
}

static int launch_fewout(const ConvP& p, hipStream_t st) {
  int pi = -1;
  if (prof_on()) {
    const double px = (double)p.n * p.hout * p.wout;
    pi = prof_begin(4, 2.0 * px * p.cout * p.cin * 9,
                    4.0 * ((double)p.n * p.cin * p.hin * p.win + (double)p.cin * 9 * p.cout + px * p.cout), st);
  }
  ConvP q = p;
  q.tiles_x = p.wout / TW;
  q.tiles_y = p.hout / FO_TH;
  hipLaunchKernelGGL(conv_fewout_kernel, dim3(q.tiles_x * q.tiles_y * q.n), dim3(256), 0, st, q);
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

static int launch_direct(const ConvP& p, int ks, int stride, int gm, hipStream_t st) {
  const int npix = p.hout * p.wout;
  int pi = -1;
  if (prof_on()) {
    const double px = (double)p.n * npix;
    pi = prof_begin(4, 2.0 * px * p.cout * p.cin * ks * ks,
                    4.0 * ((double)p.n * p.cin * p.hin * p.win + (double)p.cin * ks * ks * p.cout + px * p.cout), st);
  }
  if (p.cout <= 4) {
    dim3 grid(cdiv(npix, 256), cdiv(p.cout, 4), p.n);
    hipLaunchKernelGGL(conv_direct_kernel<4>, grid, dim3(256), 0, st, p, ks, stride, gm);
  } else {
    dim3 grid(cdiv(npix, 256), cdiv(p.cout, 8), p.n);
    hipLaunchKernelGGL(conv_direct_kernel<8>, grid, dim3(256), 0, st, p, ks, stride, gm);
  }
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

int conv2d_fwd_impl(const dsg_conv_args* a, hipStream_t st, int force_direct) {
  DSG_CHECK_ARG(a != nullptr, "dsg_conv2d_fwd: args is NULL");
  DSG_CHECK_ARG(a->src0 && a->dst, "dsg_conv2d_fwd: src0/dst must be non-NULL");
  DSG_CHECK_ARG(a->c0 > 0 && a->c1 >= 0 && a->n > 0 && a->hin > 0 && a->win > 0 && a->cout > 0,
                "dsg_conv2d_fwd: non-positive dimension");
  DSG_CHECK_ARG((a->c1 == 0) == (a->src1 == nullptr), "dsg_conv2d_fwd: src1/c1 mismatch");
  DSG_CHECK_ARG(a->ksize == 3 || a->ksize == 1, "dsg_conv2d_fwd: ksize must be 1 or 3 (got %d)", a->ksize);
  DSG_CHECK_ARG(a->stride == 1 || a->stride == 2, "dsg_conv2d_fwd: stride must be 1 or 2 (got %d)", a->stride);
  DSG_CHECK_ARG(a->upsample >= 0 && a->upsample <= 2, "dsg_conv2d_fwd: upsample must be 0, 1 or 2");
  DSG_CHECK_ARG(!(a->upsample && a->stride != 1), "dsg_conv2d_fwd: upsample requires stride 1");
  DSG_CHECK_ARG(!(a->temb && a->temb_stride <= 0), "dsg_conv2d_fwd: temb_stride must be positive");
  DSG_CHECK_ARG(a->weight_cout_stride == 0 || a->weight_cout_stride >= a->cout,
                "dsg_conv2d_fwd: weight_cout_stride smaller than cout");
  DSG_CHECK_ARG(!(a->pool2 && (a->temb || a->stride != 1)), "dsg_conv2d_fwd: pool2 excludes temb / stride 2");

  ConvP p;
  p.src0 = a->src0; p.src1 = a->src1; p.c0 = a->c0; p.c1 = a->c1; p.cin = a->c0 + a->c1;
  p.n = a->n; p.hin = a->hin; p.win = a->win;
  p.hc = a->upsample ? 2 * a->hin : a->hin;
  p.wc = a->upsample ? 2 * a->win : a->win;
  const int pad = a->ksize / 2;
  p.hout = (p.hc + 2 * pad - a->ksize) / a->stride + 1;
  p.wout = (p.wc + 2 * pad - a->ksize) / a->stride + 1;
  p.cout = a->cout; p.wstride = a->weight_cout_stride ? a->weight_cout_stride : a->cout;
  p.w = a->weight; p.bias = a->bias; p.ss = a->gn_scale_shift; p.silu = a->silu;
  p.temb = a->temb; p.temb_stride = a->temb_stride; p.res = a->residual; p.dst = a->dst;
  p.pool = a->pool2;
  p.sblk = a->src_layout; p.dblk = a->dst_layout;
  p.dt = a->compute_dtype;
  DSG_CHECK_ARG(a->compute_dtype >= DSG_F32 && a->compute_dtype <= DSG_F16,
                "dsg_conv2d_fwd: compute_dtype must be DSG_F32, DSG_BF16 or DSG_F16 (got %d)", a->compute_dtype);
  // mixed-precision modes: the channel-blocked tensors are 16-bit, [N,C,H,W] tensors fp32
  const int io16 = a->compute_dtype ? ((a->src_layout ? 1 : 0) | (a->dst_layout ? 2 : 0)) : 0;
  DSG_CHECK_ARG((a->src_layout | a->dst_layout) >= 0 && (a->src_layout | a->dst_layout) <= 1,
                "dsg_conv2d_fwd: src_layout / dst_layout must be 0 or 1");
  DSG_CHECK_ARG(!a->src_layout || (a->c0 % 8 == 0 && a->c1 % 8 == 0),
                "dsg_conv2d_fwd: channel-blocked sources need c0 %% 8 == 0 and c1 %% 8 == 0");
  DSG_CHECK_ARG(!a->dst_layout || (a->cout % 8 == 0 && !a->pool2),
                "dsg_conv2d_fwd: a channel-blocked dst needs cout %% 8 == 0 and no pool2");
  p.tiles_x = (p.wout + TW - 1) / TW; p.tiles_y = p.hout / TH;
  DSG_CHECK_ARG(!(p.pool && ((p.hout | p.wout) & 1)), "dsg_conv2d_fwd: pool2 needs even output dims");

  if (a->src_operand != nullptr)  // pre-staged operand image: only the split-path kernels that DMA it serve the call
    DSG_CHECK_SHAPE(!force_direct && conv_h2_takes_operand(a, p.hout, p.wout, false),
                    "dsg_conv2d_fwd: this call's kernel stages its own patch (dsg_conv2d_takes_operand reports 0): pass "
                    "src_operand = NULL");
  if (a->gnb_x0 != nullptr) {  // GroupNorm-backward statistics: only the split-path kernels with that epilogue serve the call
    DSG_CHECK_ARG(a->gnb_ss != nullptr && a->stats_out != nullptr && (a->gnb_x1 == nullptr || a->gnb_c0 > 0),
                  "dsg_conv2d_fwd: gnb_x0 needs gnb_ss and stats_out (and gnb_c0 with gnb_x1)");
    DSG_CHECK_SHAPE(!force_direct && conv_h2_gnb_ok(a, p.hout, p.wout),
                    "dsg_conv2d_fwd: this call's kernel has no GroupNorm-backward epilogue (dsg_conv2d_gnb_supported reports 0): "
                    "pass gnb_x0 = NULL and run the statistics pass");
    return conv_h2_launch(a, p.hout, p.wout, st);
  }
  if (a->s2_window4) {  // data gradient of an up-sampler conv as one 4x4 stride-2 window: the space-to-depth kernel only
    DSG_CHECK_ARG(a->s2_window4 == 1 && a->stride == 2 && a->ksize == 3 && a->weight_h2_s2 != nullptr,
                  "dsg_conv2d_fwd: s2_window4 must be 0 or 1 and needs stride 2, ksize 3 and weight_h2_s2 (pack kind 5)");
    DSG_CHECK_SHAPE(!force_direct && conv_h2_s2(a, p.hout, p.wout),
                    "dsg_conv2d_fwd: s2_window4: the space-to-depth kernel does not take this call (channel-blocked tensors, "
                    "c0 %% 8 == 0, cout %% 8 == 0, output rows %% 8 == 0, output width 8, 16 or a multiple of 32)");
    return conv_h2_launch(a, p.hout, p.wout, st);
  }
  if (a->sc_weight_h2 != nullptr) {  // fused shortcut: only the split-path kernel that contracts it serves the call
    DSG_CHECK_ARG(a->sc_src0 != nullptr && a->sc_c0 > 0 && a->sc_c1 >= 0 && (a->sc_c1 == 0) == (a->sc_src1 == nullptr) &&
                      a->residual == nullptr,
                  "dsg_conv2d_fwd: sc_weight_h2 needs sc_src0 / sc_c0 (sc_src1 / sc_c1 together) and no residual");
    DSG_CHECK_SHAPE(!force_direct && conv_h2_sc_fusable(a, p.hout, p.wout),
                    "dsg_conv2d_fwd: this call cannot fuse a shortcut (dsg_conv2d_fuses_shortcut reports 0): run the 1x1 "
                    "conv on its own and pass its result as residual");
    return conv_h2_launch(a, p.hout, p.wout, st);
  }
  // (kernel selection never depends on whether `weight` was passed: conv_in / conv_out READ the fp32 engine layout, so a
  //  call without it is refused -- with the message TrainState.lazy_w retries on -- instead of silently taking another kernel)
  if (!force_direct && (conv_in_eligible(a, p.hout, p.wout) || conv_out_eligible(a, p.hout, p.wout))) {
    DSG_CHECK_ARG(a->weight != nullptr,
                  "dsg_conv2d_fwd: weight is NULL and this call is served by the conv_in / conv_out kernel (needs the fp32 engine layout)");
    return conv_in_eligible(a, p.hout, p.wout) ? conv_in_launch(a, p.hout, p.wout, st) : conv_out_launch(a, p.hout, p.wout, st);
  }
  if (!force_direct && conv_h2_eligible(a, p.hout, p.wout)) return conv_h2_launch(a, p.hout, p.wout, st);
  // (`weight`, the fp32 engine layout, may be NULL for a call the operand-image kernels serve: a training step re-lays
  // out every weight it passes here, and most calls never read it)
  DSG_CHECK_ARG(a->weight != nullptr,
                "dsg_conv2d_fwd: weight is NULL and this call is not served by the weight_h2* kernels (needs the fp32 engine layout)");
  DSG_CHECK_ARG(a->stats_out == nullptr,
                "dsg_conv2d_fwd: stats_out given but this call is not served by the kernel that produces them "
                "(dsg_conv2d_stats_tiles reports 0)");

  const bool narrow = (p.wout == 16 || p.wout == 8) && !p.pool;  // less than one tile wide: the spare lanes idle
  const bool tile_ok = (p.wout % TW == 0 || narrow) && (p.hout % TH == 0) && (p.wstride % 32 == 0) && p.cin <= 2048 &&
                       (p.wstride >= ((p.cout + 31) / 32) * 32);
  const int s = a->stride, k = a->ksize, u = a->upsample;
  if (!force_direct && g_conv_fewout && k == 3 && s == 1 && u == 0 && !p.pool && p.cout <= 4 && p.cin % FO_KC == 0 &&
      (p.c1 == 0 || p.c0 % FO_KC == 0) && p.wout % TW == 0 && p.hout % FO_TH == 0)
    return launch_fewout(p, st);  // conv_out: too few output channels for the matrix cores
  if (io16 == 2 && !force_direct && tile_ok && k == 3 && s == 1 && u == 0 && !p.pool) {
    // conv_in of the mixed-precision modes: fp32 [N,C,H,W] image -> 16-bit channel-blocked activations
    const bool mt2 = (p.wstride % 64 == 0) && p.cout > 32;
    const bool kc4 = p.cin <= 4 || !(p.c1 == 0 || p.c0 % 8 == 0);
    if (kc4 && (p.c1 == 0 || p.c0 % 4 == 0))
      return mt2 ? launch_mfma<3, 1, 0, 2, 4, 2>(p, st) : launch_mfma<3, 1, 0, 1, 4, 2>(p, st);
    if (p.c1 == 0 || p.c0 % 8 == 0)
      return mt2 ? launch_mfma<3, 1, 0, 2, 8, 2>(p, st) : launch_mfma<3, 1, 0, 1, 8, 2>(p, st);
  }
  if (io16 == 3 && !force_direct && tile_ok && k == 3 && s == 2 && u == 0 && !p.pool && (p.c1 == 0 || p.c0 % 4 == 0)) {
    // down-sampler convs whose result is narrower than a tile (16 x 16 and below): the exact fp32 chain on 16-bit tensors
    const bool mt2 = (p.wstride % 64 == 0) && p.cout > 32;
    return mt2 ? launch_mfma<3, 2, 0, 2, 4, 3>(p, st) : launch_mfma<3, 2, 0, 1, 4, 3>(p, st);
  }
  if (io16 == 3 && !force_direct && tile_ok && k == 3 && s == 1 && u == 1 && !p.pool && (p.c1 == 0 || p.c0 % 8 == 0)) {
    // up-sampler convs of maps the folded split kernel does not tile (source narrower than 32 columns)
    const bool mt2 = (p.wstride % 64 == 0) && p.cout > 32;
    return mt2 ? launch_mfma<3, 1, 1, 2, 8, 3>(p, st) : launch_mfma<3, 1, 1, 1, 8, 3>(p, st);
  }
  DSG_CHECK_SHAPE(io16 == 0,
                  "dsg_conv2d_fwd: no %s kernel serves this call with channel-blocked tensors (k %d, stride %d, "
                  "upsample %d, cin %d, cout %d, %dx%d); the 16-bit modes take the shapes of dsg_conv2d_fwd's "
                  "matrix-core path, conv_in and conv_out only", a->compute_dtype == DSG_BF16 ? "bf16" : "fp16", k, s, u,
                  p.cin, p.cout, p.hout, p.wout);
  if (!force_direct && tile_ok) {
    const bool mt2 = (p.wstride % 64 == 0) && p.cout > 32;
    const bool dual_ok4 = p.c1 == 0 || p.c0 % 4 == 0;
    const bool dual_ok8 = p.c1 == 0 || p.c0 % 8 == 0;
    if (k == 3 && s == 1) {
      // KC=4 keeps 3 workgroups per CU (30 KB LDS, 127 VGPRs): better when the grid has >= 3 per CU to give;
      // KC=8 (2 per CU, half the barriers) wins on the low-resolution levels.  Few channels: KC=4.
      const int nblk = p.tiles_x * p.tiles_y * p.n * ((p.cout + 63) / 64);
      const bool kc4 = p.cin <= 4 || !dual_ok8 || g_conv_kc == 4 || (g_conv_kc == 0 && nblk >= 768);
      if (kc4 && dual_ok4) {
        if (u == 0) return mt2 ? launch_mfma<3, 1, 0, 2, 4>(p, st) : launch_mfma<3, 1, 0, 1, 4>(p, st);
        if (u == 1) return mt2 ? launch_mfma<3, 1, 1, 2, 4>(p, st) : launch_mfma<3, 1, 1, 1, 4>(p, st);
        return mt2 ? launch_mfma<3, 1, 2, 2, 4>(p, st) : launch_mfma<3, 1, 2, 1, 4>(p, st);
      }
      if (dual_ok8) {
        if (u == 0) return mt2 ? launch_mfma<3, 1, 0, 2, 8>(p, st) : launch_mfma<3, 1, 0, 1, 8>(p, st);
        if (u == 1) return mt2 ? launch_mfma<3, 1, 1, 2, 8>(p, st) : launch_mfma<3, 1, 1, 1, 8>(p, st);
        return mt2 ? launch_mfma<3, 1, 2, 2, 8>(p, st) : launch_mfma<3, 1, 2, 1, 8>(p, st);
      }
    }
    if (k == 3 && s == 2 && u == 0 && dual_ok4)
      return mt2 ? launch_mfma<3, 2, 0, 2, 4>(p, st) : launch_mfma<3, 2, 0, 1, 4>(p, st);
    if (k == 1 && s == 1 && u == 0 && (p.c1 == 0 || p.c0 % 16 == 0))
      return mt2 ? launch_mfma<1, 1, 0, 2, 16>(p, st) : launch_mfma<1, 1, 0, 1, 16>(p, st);
  }
  DSG_CHECK_SHAPE(!p.pool, "dsg_conv2d_fwd: pool2 is only implemented on the MFMA path (shape %dx%d, cin %d)",
                  p.hout, p.wout, p.cin);
  DSG_CHECK_SHAPE(!p.sblk && !p.dblk,
                  "dsg_conv2d_fwd: channel-blocked tensors are only taken by the matrix-core and conv_out kernels "
                  "(shape %dx%d, cin %d)", p.hout, p.wout, p.cin);
  return launch_direct(p, k, s, u, st);
}

}  // namespace dsg

namespace dsg { int conv_h2_tuning_epoch(); }
static int g_tuning_epoch = 0;   // every accepted dsg_set_tuning call (whatever file owns the switch)
DSG_API int32_t dsg_tuning_epoch(void) { return dsg::conv_h2_tuning_epoch() + g_tuning_epoch; }

// Tuning / A-B switches (key 1: K-chunk of the fp32 3x3 kernel, 0 = auto | 4 | 8; key 2: fp16x2-split 3x3 kernel
// on/off).  Not part of the reference surface.
static int set_tuning_impl(int32_t key, int32_t value);
DSG_API int dsg_set_tuning(int32_t key, int32_t value) {
  const int rc = set_tuning_impl(key, value);
  if (rc == DSG_OK) ++g_tuning_epoch;   // host-side caches keyed on dsg_tuning_epoch see EVERY accepted key
  return rc;
}
static int set_tuning_impl(int32_t key, int32_t value) {
  {  // a test / measurement hook: production processes keep the library's global state immutable
    const char* t = getenv("DSG_TESTING");
    if (t == nullptr || t[0] != '1')
      return dsg::fail(DSG_ERR_INVALID_ARG, "dsg_set_tuning: kernel-selection switches are a test hook (set DSG_TESTING=1 in the "
                                            "environment); per-plan choices are in dsg_unet_config.flags");
  }
  if (key == 1 && (value == 0 || value == 4 || value == 8)) {
    dsg::g_conv_kc = value;
    return DSG_OK;
  }
  if (key == 2 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_enabled(value);
    return DSG_OK;
  }
  if (key == 3 && (value == 0 || value == 2 || value == 3 || value == 4)) {
    dsg::conv_h2_set_rows(value);
    return DSG_OK;
  }
  if (key == 10 && (value == 0 || value == 1)) {
    dsg::g_conv_fewout = value;
    return DSG_OK;
  }
  if (key == 8 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_fold(value);
    return DSG_OK;
  }
  if (key == 7 && (value == 0 || value == 1)) {
    dsg::wgrad_h2_set_enabled(value);
    return DSG_OK;
  }
  if (key == 6 && (value == 4 || value == 8)) {
    dsg::conv_h2_set_waves(value);
    return DSG_OK;
  }
  if (key == 17 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_bm32_small(value);
    return DSG_OK;
  }
  if (key == 19 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_splitk(value);
    return DSG_OK;
  }
  if (key == 20 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_ws2(value);
    return DSG_OK;
  }
  if (key == 25 && (value == 0 || value == 1)) {
    dsg::attention_set_blocked(value);
    dsg::conv_h2_set_fuse_sc(dsg::conv_h2_get_fuse_sc());  // (bumps the tuning epoch: the plan's arena changes)
    return DSG_OK;
  }
  if (key == 23 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_fuse_sc(value);
    return DSG_OK;
  }
  if (key == 26 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_pre(value);
    return DSG_OK;
  }
  if (key == 34 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_splitk_mid(value);
    return DSG_OK;
  }
  if (key == 32 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_narrow(value);
    return DSG_OK;
  }
  if (key == 36 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_rows_rule(value);
    return DSG_OK;
  }
  if (key == 37 && value >= 0 && value <= 3) {
    dsg::conv_h2_set_gnb(value);
    return DSG_OK;
  }
  if (key == 38 && (value == 0 || value == 1)) {
    dsg::attention_set_bwd_split(value);
    return DSG_OK;
  }
  if (key == 41 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_gnb_bm64(value);
    return DSG_OK;
  }
  if (key == 40 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_s2_nchw(value);
    return DSG_OK;
  }
  if (key == 39 && (value == 0 || value == 1)) {
    dsg::conv_wgrad16_set_fold(value);
    return DSG_OK;
  }
  if (key == 31 && (value == 0 || value == 1)) {
    dsg::wgrad_h2_set_wide(value);
    return DSG_OK;
  }
  if (key == 30 && (value == 0 || value == 1)) {
    dsg::conv_wgrad16_set_pw(value);
    return DSG_OK;
  }
  if (key == 29 && (value == 0 || value == 1)) {
    dsg::conv_wgrad16_set_wide(value);
    return DSG_OK;
  }
  if (key == 27 && value >= 1) {
    dsg::conv_h2_set_pre_min_ct(value);
    return DSG_OK;
  }
  if (key == 18 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_bm128(value);
    return DSG_OK;
  }
  if (key == 16 && value >= 0) {
    dsg::conv_h2_set_bm32(value);
    return DSG_OK;
  }
  if (key == 15 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_s2(value);
    return DSG_OK;
  }
  if (key == 14 && (value == 0 || value == 1)) {
    dsg::attention_set_mfma(value);
    return DSG_OK;
  }
  if (key == 13 && (value == 0 || value == 1)) {
    dsg::unet_set_blocked(value);
    return DSG_OK;
  }
  if (key == 11 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_pw_occ2(value);
    return DSG_OK;
  }
  if (key == 22 && (value == 0 || value == 1)) {
    dsg::conv_out_set_enabled(value);
    return DSG_OK;
  }
  if (key == 21 && (value == 0 || value == 1)) {
    dsg::conv_in_set_enabled(value);
    return DSG_OK;
  }
  if (key == 5 && (value == 0 || value == 1)) {
    dsg::conv_h2_set_stats(value);
    return DSG_OK;
  }
  return dsg::fail(DSG_ERR_INVALID_ARG, "dsg_set_tuning: unknown key/value %d/%d", key, value);
}

DSG_API int dsg_conv2d_stats_tiles(const dsg_conv_args* a, int32_t* tiles) {
  DSG_CHECK_ARG(a != nullptr && tiles != nullptr, "dsg_conv2d_stats_tiles: NULL pointer");
  DSG_CHECK_ARG(a->ksize == 3 || a->ksize == 1, "dsg_conv2d_stats_tiles: ksize must be 1 or 3");
  DSG_CHECK_ARG(a->stride == 1 || a->stride == 2, "dsg_conv2d_stats_tiles: stride must be 1 or 2");
  const int hc = a->upsample ? 2 * a->hin : a->hin, wc = a->upsample ? 2 * a->win : a->win;
  const int pad = a->ksize / 2;
  const int hout = (hc + 2 * pad - a->ksize) / a->stride + 1, wout = (wc + 2 * pad - a->ksize) / a->stride + 1;
  *tiles = dsg::conv_in_eligible(a, hout, wout) ? dsg::conv_in_stats_tiles(a, hout, wout) : dsg::conv_h2_stats_tiles(a, hout, wout);
  return DSG_OK;
}

DSG_API int dsg_conv2d_splitk_bytes(const dsg_conv_args* a, size_t* bytes) {
  DSG_CHECK_ARG(a != nullptr && bytes != nullptr, "dsg_conv2d_splitk_bytes: NULL pointer");
  DSG_CHECK_ARG((a->ksize == 3 || a->ksize == 1) && (a->stride == 1 || a->stride == 2), "dsg_conv2d_splitk_bytes: bad ksize / stride");
  const int hc = a->upsample ? 2 * a->hin : a->hin, wc = a->upsample ? 2 * a->win : a->win;
  const int pad = a->ksize / 2;
  const int hout = (hc + 2 * pad - a->ksize) / a->stride + 1, wout = (wc + 2 * pad - a->ksize) / a->stride + 1;
  const int slices = dsg::conv_h2_splitk_slices(a, hout, wout, nullptr);
  *bytes = slices > 1 ? (size_t)slices * a->n * a->cout * hout * wout * sizeof(float) : 0;
  return DSG_OK;
}

DSG_API int dsg_conv2d_fuses_shortcut(const dsg_conv_args* a, int32_t* yes) {
  DSG_CHECK_ARG(a != nullptr && yes != nullptr, "dsg_conv2d_fuses_shortcut: NULL pointer");
  *yes = 0;
  if (a->sc_weight_h2 == nullptr || a->sc_src0 == nullptr || a->sc_c0 <= 0 || a->ksize != 3 || a->stride != 1 || a->upsample)
    return DSG_OK;
  *yes = dsg::conv_h2_sc_fusable(a, a->hin, a->win) ? 1 : 0;
  return DSG_OK;
}

DSG_API int dsg_conv2d_gnb_supported(const dsg_conv_args* a, int32_t* yes) {
  DSG_CHECK_ARG(a != nullptr && yes != nullptr, "dsg_conv2d_gnb_supported: NULL pointer");
  *yes = 0;
  if (a->gnb_x0 == nullptr || a->ksize != 3 || a->stride != 1 || a->upsample) return DSG_OK;
  if (a->splitk_ws != nullptr && a->compute_dtype == DSG_F32 && a->src_layout == 1 && a->dst_layout == 1) return DSG_OK;
  *yes = dsg::conv_h2_gnb_ok(a, a->hin, a->win) ? 1 : 0;
  return DSG_OK;
}

DSG_API int dsg_conv2d_takes_operand(const dsg_conv_args* a, int32_t* yes) {
  DSG_CHECK_ARG(a != nullptr && yes != nullptr, "dsg_conv2d_takes_operand: NULL pointer");
  *yes = 0;
  if (a->ksize != 3 || a->stride != 1 || a->upsample < 0 || a->upsample > 1) return DSG_OK;
  const int hout = a->upsample ? 2 * a->hin : a->hin, wout = a->upsample ? 2 * a->win : a->win;
  *yes = dsg::conv_h2_takes_operand(a, hout, wout, true) ? 1 : 0;
  return DSG_OK;
}

DSG_API int dsg_conv2d_fwd(const dsg_conv_args* a, void* stream) {
  return dsg::conv2d_fwd_impl(a, static_cast<hipStream_t>(stream), 0);
}

// Test hook: same contract, forced through the VALU reference kernel (cross-check of the MFMA path).
DSG_API int dsg_conv2d_fwd_direct(const dsg_conv_args* a, void* stream) {
  return dsg::conv2d_fwd_impl(a, static_cast<hipStream_t>(stream), 1);
}

static int relayout(const float* w_oihw, float* dst, int32_t cout, int32_t cin, int32_t ksize, int32_t cout_total,
                    int32_t cout_off, int mode, void* stream) {
  DSG_CHECK_ARG(w_oihw && dst, "dsg_conv_weight_relayout: NULL pointer");
  DSG_CHECK_ARG(cout > 0 && cin > 0 && (ksize == 1 || ksize == 3), "dsg_conv_weight_relayout: bad dims");
  const int width = mode == 0 ? cout : cin;
  DSG_CHECK_ARG(cout_off >= 0 && cout_off + width <= cout_total, "dsg_conv_weight_relayout: column range");
  const int64_t total = (int64_t)cout * cin * ksize * ksize;
  const int blocks = (int)std::min<int64_t>(dsg::cdiv64(total, 256), 4096);
  hipLaunchKernelGGL(dsg::weight_relayout_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w_oihw, dst, cout, cin, ksize * ksize, cout_total, cout_off, mode);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_conv_weight_relayout(const float* w_oihw, float* dst, int32_t cout, int32_t cin, int32_t ksize,
                                     int32_t cout_total, int32_t cout_off, void* stream) {
  return relayout(w_oihw, dst, cout, cin, ksize, cout_total, cout_off, 0, stream);
}

DSG_API int dsg_conv_weight_relayout_dgrad(const float* w_oihw, float* dst, int32_t cout, int32_t cin, int32_t ksize,
                                           int32_t cin_total, int32_t cin_off, void* stream) {
  return relayout(w_oihw, dst, cout, cin, ksize, cin_total, cin_off, 1, stream);
}
