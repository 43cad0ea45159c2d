// Weight-gradient convolution for gfx950 (MI355X), fp32 on the f32-input matrix cores.
//
// Backward of every nn.Conv2d / attention nn.Linear of diffusers' UNet2DModel as trained by
// DriveSceneGen (reference: DriveSceneGen/pipeline/training_pipeline.py:84-86 -- forward, mse_loss,
// accelerator.backward(loss)).  The data gradient reuses conv.hip with transposed/flipped weights
// (dsg_conv_weight_relayout_dgrad); this file is the other half:
//
//   dW[co][ci][tap] += sum_{n, pixel} dY[n][co][pixel] * A[n][ci][pixel*stride + tap]
//
// where A = act(affine(cat(src0, src1))) is RECOMPUTED in the gather from the saved pre-norm tensor and
// the GroupNorm scale/shift (nothing but conv outputs is kept for backward).
//
// GEMM view per tap:  D[ci][co] = sum_pixels A[ci][pixel+tap] * dY[co][pixel]   (K = N*H*W, split-K)
//   A operand = patch (M = ci), B operand = dY (N = co); a wave owns one 32ci x 32co tile with all
//   kh*kw tap accumulators (9 x 16 registers), so one dY operand read feeds 9 MFMAs.
// A workgroup (4 waves) owns (32*CIT ci) x 64 co and walks its share of 2-row x 32-col pixel tiles of all
// images (grid.y = K split); partial sums are added to dW (OIHW, the checkpoint layout) with fp32 atomics.
// LDS: dY tile [64 co][64 px (+1 pad)], patch [ci][rows*cols (odd stride)] -- both conflict-free for the
// lane patterns of v_mfma_f32_32x32x2_f32 (A/B: 32 consecutive rows at a fixed k).
#include "dsg_common.h"
#include <algorithm>

namespace dsg {

bool prof_on();
int prof_begin(int kid, double flops, double bytes, hipStream_t st);
void prof_end(int idx, hipStream_t st);

struct WgradP {
  const float* src0;
  const float* src1;
  int c0, c1, cin;
  int n, hin, win;  // source dims
  int hc, wc;       // conv-input dims (after optional nearest upsample)
  int hout, wout;
  int cout;
  const float* dy;  // [N, dy_ctotal, hout, wout]; this conv's channels start at dy_coff
  int dy_ctotal, dy_coff;
  const float* ss;  // optional [N][cin][2]
  int silu;
  float* dw;  // [cout][cin][taps], accumulated
  float* ws;  // split-K partials [nsplit][taps][cin_pad][cout_pad] (MFMA path)
  int cin_pad, cout_pad;
  int tiles_x, tiles_y, ntiles;
  int ci_blocks;
};

__device__ __forceinline__ float silu_fast_b(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

constexpr int WG_SR = 2;    // output rows per stage
constexpr int WG_CO = 64;   // couts per workgroup
constexpr int WG_DYS = 65;  // padded row stride of the dY tile

template <int KS, int STRIDE, int CIT>
struct WgradGeom {
  static constexpr int TAPS = KS * KS;
  static constexpr int PH = (WG_SR - 1) * STRIDE + KS;
  static constexpr int PW = 31 * STRIDE + KS;
  static constexpr int PSZ = PH * PW;
  static constexpr int PST = (PSZ & 1) ? PSZ : PSZ + 1;  // odd channel stride: conflict-free A-operand reads
  static constexpr int CIB = 32 * CIT;
  static constexpr int LDS_FLOATS = WG_CO * WG_DYS + CIB * PST;
};

template <int KS, int STRIDE, int UPS, int CIT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradP p) {
  using G = WgradGeom<KS, STRIDE, CIT>;
  constexpr int TAPS = G::TAPS, PW = G::PW, PSZ = G::PSZ, PST = G::PST, CIB = G::CIB;
  constexpr int TPC = 256 / CIB;              // threads per patch channel
  constexpr int NE = (PSZ + TPC - 1) / TPC;   // patch elements per thread per stage
  static_assert(NE <= 64, "validity mask is 64 bits");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* DYl = smem;                   // [64 co][65]
  float* Al = smem + WG_CO * WG_DYS;   // [CIB][PST]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  // wave -> (ci tile, co tile): CIT=2: 2x2; CIT=1: 1 ci tile x 2 co tiles, waves 2,3 take the odd pixel half
  const int cit = (CIT == 2) ? (wave >> 1) : 0;
  const int cot = wave & 1;
  const int khalf = (CIT == 2) ? 0 : (wave >> 1);  // CIT=1: waves split the stage's 32 k-steps in two

  const int cib = blockIdx.x % p.ci_blocks;
  const int cob = blockIdx.x / p.ci_blocks;
  const int ci0 = cib * CIB, co0 = cob * WG_CO;
  const int plane = p.hin * p.win;
  const int oplane = p.hout * p.wout;

  // patch staging: this thread owns channel tc of the block, NE positions
  const int tc = tid / TPC;
  const int tr = tid - tc * TPC;
  const int gci = ci0 + tc;
  const bool cok = gci < p.cin;
  const int gcc = min(gci, p.cin - 1);
  const bool from0 = gcc < p.c0;
  const bool has_ss = p.ss != nullptr;
  const bool do_silu = has_ss && p.silu;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float xr[NE];
  float4 dr[4];

  // global -> registers for one pixel tile (issued one stage ahead of its use)
  float sc = 1.f, sh = 0.f;
  unsigned long long valid = 0;  // NE can exceed 32 (34 for 3x3, 41 for stride 2)
  auto load_tile = [&](int t) {
    int tt = t;
    const int tx = tt % p.tiles_x;
    tt /= p.tiles_x;
    const int ty = tt % p.tiles_y;
    const int n = tt / p.tiles_y;
    const int oy0 = ty * WG_SR, ox0 = tx * 32;
    const int iy0 = oy0 * STRIDE - KS / 2, ix0 = ox0 * STRIDE - KS / 2;
    const float* sp = from0 ? p.src0 + ((size_t)n * p.c0 + gcc) * plane
                            : p.src1 + ((size_t)n * p.c1 + (gcc - p.c0)) * plane;
    if (has_ss) {
      const float2 s2 = *reinterpret_cast<const float2*>(p.ss + ((size_t)n * p.cin + gcc) * 2);
      sc = s2.x;
      sh = s2.y;
    }
    valid = 0;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int r = tr + TPC * i;
      float v = 0.f;
      if (r < PSZ) {
        const int py = r / PW, px = r - py * PW;
        const int gy = iy0 + py, gx = ix0 + px;
        if (gy >= 0 && gy < p.hc && gx >= 0 && gx < p.wc) {
          const int sy = UPS ? (gy >> 1) : gy, sx = UPS ? (gx >> 1) : gx;
          v = sp[sy * p.win + sx];
          valid |= 1ull << i;
        }
      }
      xr[i] = v;
    }
    // dY tile: 64 co x (2 rows x 32 px): 1024 float4, 4 per thread
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;  // co = idx/16, row = (idx/8)&1, q4 = idx&7
      const int co = idx >> 4, row = (idx >> 3) & 1, q4 = idx & 7;
      const int gco = co0 + co;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gco < p.cout)
        v = *reinterpret_cast<const float4*>(p.dy + ((size_t)n * p.dy_ctotal + p.dy_coff + gco) * oplane +
                                             (size_t)(oy0 + row) * p.wout + ox0 + q4 * 4);
      dr[i] = v;
    }
  };

  if ((int)blockIdx.y < p.ntiles) load_tile(blockIdx.y);
  for (int t = blockIdx.y; t < p.ntiles; t += gridDim.y) {
    __syncthreads();  // previous stage's MFMA phase is done with LDS
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int r = tr + TPC * i;
      if (r < PSZ) {
        float v = xr[i] * sc + sh;
        if (do_silu) v = silu_fast_b(v);
        Al[tc * PST + r] = (((valid >> i) & 1ull) && cok) ? v : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      const int co = idx >> 4, row = (idx >> 3) & 1, q4 = idx & 7;
      float* d = DYl + co * WG_DYS + row * 32 + q4 * 4;
      d[0] = dr[i].x; d[1] = dr[i].y; d[2] = dr[i].z; d[3] = dr[i].w;
    }
    __syncthreads();
    if (t + (int)gridDim.y < p.ntiles) load_tile(t + gridDim.y);  // next tile's loads fly under the MFMAs

    // ---- MFMA phase: 32 k-steps (pixel pairs) x TAPS ----
    const float* al = Al + (cit * 32 + l31) * PST;
    const float* dl = DYl + (cot * 32 + l31) * WG_DYS;
    constexpr int KSTEPS = (CIT == 2) ? 32 : 16;
#pragma unroll 4
    for (int ks0 = 0; ks0 < KSTEPS; ++ks0) {
      const int ks = ks0 + khalf * 16;
      const int px = 2 * ks + half;     // output pixel within the 2x32 stage
      const int row = px >> 5, col = px & 31;
      const float b = dl[px];
      const float* ap = al + (row * STRIDE) * PW + col * STRIDE;
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp) {
        const float a = ap[(tp / KS) * PW + (tp % KS)];
        acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tp], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: D[i = ci][j = co]; lane holds co = l31, ci rows (r&3) + 8*(r>>2) + 4*half ----
  // Partials go to the workspace slab of this K-split ([tap][ci][co], co contiguous -> 128-B stores); a second
  // kernel sums the splits in a fixed order into dW (deterministic, and ~100x cheaper than 37 M fp32 atomics).
  // CIT == 1 splits the k-steps over two wave pairs: the second pair adds into a second slab half.
  const int co = co0 + cot * 32 + l31;
  float* wsb = p.ws + ((size_t)blockIdx.y * (CIT == 2 ? 1 : 2) + khalf) * TAPS * p.cin_pad * p.cout_pad;
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + cit * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      wsb[((size_t)tp * p.cin_pad + ci) * p.cout_pad + co] = acc[tp][r];
    }
  }
}

// dw[co][ci][tp] += sum_s ws[s][tp][ci][co]; one thread per (tp, ci, co), co fastest (coalesced reads)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int nslab, int taps, int cin,
                                                           int cout, int cin_pad, int cout_pad,
                                                           float* __restrict__ dw) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  const int64_t slab = (int64_t)taps * cin_pad * cout_pad;
  if (i >= slab) return;
  const int co = (int)(i % cout_pad);
  const int ci = (int)((i / cout_pad) % cin_pad);
  const int tp = (int)(i / ((int64_t)cout_pad * cin_pad));
  if (co >= cout || ci >= cin) return;
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += ws[k * slab + i];
  dw[((size_t)co * cin + ci) * taps + tp] += s;
}

// Generic VALU fallback (odd spatial sizes): one thread per (co, ci, tap), loops over all pixels.
__global__ __launch_bounds__(256) void conv_wgrad_direct_kernel(WgradP p, int ks, int stride, int ups) {
  const int taps = ks * ks;
  const int64_t total = (int64_t)p.cout * p.cin * taps;
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= total) return;
  const int tp = (int)(i % taps);
  const int ci = (int)((i / taps) % p.cin);
  const int co = (int)(i / ((int64_t)taps * p.cin));
  const int dyk = tp / ks, dxk = tp % ks, pad = ks / 2;
  const int plane = p.hin * p.win;
  float s = 0.f;
  for (int n = 0; n < p.n; ++n) {
    const float* sp = (ci < p.c0) ? p.src0 + ((size_t)n * p.c0 + ci) * plane
                                  : p.src1 + ((size_t)n * p.c1 + (ci - p.c0)) * plane;
    float sc = 1.f, sh = 0.f;
    if (p.ss) {
      sc = p.ss[((size_t)n * p.cin + ci) * 2];
      sh = p.ss[((size_t)n * p.cin + ci) * 2 + 1];
    }
    const float* dp = p.dy + ((size_t)n * p.dy_ctotal + p.dy_coff + co) * p.hout * p.wout;
    for (int oy = 0; oy < p.hout; ++oy) {
      const int gy = oy * stride - pad + dyk;
      if (gy < 0 || gy >= p.hc) continue;
      for (int ox = 0; ox < p.wout; ++ox) {
        const int gx = ox * stride - pad + dxk;
        if (gx < 0 || gx >= p.wc) continue;
        float v = sp[(ups ? gy >> 1 : gy) * p.win + (ups ? gx >> 1 : gx)];
        if (p.ss) {
          v = v * sc + sh;
          if (p.silu) v = silu_f(v);
        }
        s = fmaf(v, dp[oy * p.wout + ox], s);
      }
    }
  }
  p.dw[i] += s;
}

static int wgrad_nsplit(int pairs, int ntiles) { return std::max(1, std::min(ntiles, cdiv(1024, pairs))); }

template <int KS, int STRIDE, int UPS, int CIT>
static int launch_wgrad(WgradP p, size_t ws_bytes, hipStream_t st) {
  using G = WgradGeom<KS, STRIDE, CIT>;
  p.ci_blocks = cdiv(p.cin, G::CIB);
  const int co_blocks = cdiv(p.cout, WG_CO);
  const int pairs = p.ci_blocks * co_blocks;
  const int nsplit = wgrad_nsplit(pairs, p.ntiles);
  p.cin_pad = p.ci_blocks * G::CIB;
  p.cout_pad = co_blocks * WG_CO;
  const int nslab = nsplit * (CIT == 2 ? 1 : 2);
  const size_t need = (size_t)nslab * G::TAPS * p.cin_pad * p.cout_pad * sizeof(float);
  if (p.ws == nullptr || ws_bytes < need)
    return fail(DSG_ERR_WORKSPACE_TOO_SMALL, "dsg_conv2d_wgrad: workspace %zu bytes < required %zu", ws_bytes, need);
  const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
  auto kern = conv_wgrad_kernel<KS, STRIDE, UPS, CIT>;
  static bool raised = false;
  if (!raised) {
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    raised = true;
  }
  int pi = -1;
  if (prof_on())
    pi = prof_begin(5, 2.0 * p.n * p.hout * p.wout * (double)p.cout * p.cin * G::TAPS,
                    4.0 * ((double)p.n * p.cin * p.hin * p.win + (double)p.n * p.cout * p.hout * p.wout), st);
  hipLaunchKernelGGL(kern, dim3(pairs, nsplit), dim3(256), lds, st, p);
  DSG_LAUNCH_CHECK();
  const int64_t slab = (int64_t)G::TAPS * p.cin_pad * p.cout_pad;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(slab, 256)), dim3(256), 0, st, p.ws, nslab, G::TAPS, p.cin,
                     p.cout, p.cin_pad, p.cout_pad, p.dw);
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

// bytes of split-K workspace the MFMA path needs for these dims (0 for the VALU fallback)
static size_t wgrad_ws_bytes(int cin, int cout, int ks, int stride, int hout, int wout, int n) {
  if ((wout % 32) || (hout % WG_SR)) return 0;
  const int cit = (ks == 3 && stride == 2) ? 1 : (cin <= 32 ? 1 : 2);
  const int cib = 32 * cit;
  const int ci_blocks = cdiv(cin, cib), co_blocks = cdiv(cout, WG_CO);
  const int ntiles = (wout / 32) * (hout / WG_SR) * n;
  const int nslab = wgrad_nsplit(ci_blocks * co_blocks, ntiles) * (cit == 2 ? 1 : 2);
  return (size_t)nslab * ks * ks * ci_blocks * cib * co_blocks * WG_CO * sizeof(float);
}

}  // namespace dsg

DSG_API int dsg_conv2d_wgrad(const dsg_conv_wgrad_args* a, void* stream) {
  using namespace dsg;
  DSG_CHECK_ARG(a != nullptr, "dsg_conv2d_wgrad: args is NULL");
  DSG_CHECK_ARG(a->src0 && a->dy && a->dw, "dsg_conv2d_wgrad: src0/dy/dw must be non-NULL");
  DSG_CHECK_ARG(a->c0 > 0 && a->c1 >= 0 && a->n > 0 && a->hin > 0 && a->win > 0 && a->cout > 0,
                "dsg_conv2d_wgrad: non-positive dimension");
  DSG_CHECK_ARG((a->c1 == 0) == (a->src1 == nullptr), "dsg_conv2d_wgrad: src1/c1 mismatch");
  DSG_CHECK_ARG(a->ksize == 3 || a->ksize == 1, "dsg_conv2d_wgrad: ksize must be 1 or 3");
  DSG_CHECK_ARG(a->stride == 1 || a->stride == 2, "dsg_conv2d_wgrad: stride must be 1 or 2");
  DSG_CHECK_ARG(a->upsample == 0 || a->upsample == 1, "dsg_conv2d_wgrad: upsample must be 0 or 1");
  DSG_CHECK_ARG(!(a->upsample && a->stride != 1), "dsg_conv2d_wgrad: upsample requires stride 1");
  DSG_CHECK_ARG(a->dy_coff >= 0 && (a->dy_ctotal == 0 || a->dy_coff + a->cout <= a->dy_ctotal),
                "dsg_conv2d_wgrad: dy channel window out of range");
  hipStream_t st = static_cast<hipStream_t>(stream);
  WgradP p;
  p.src0 = a->src0; p.src1 = a->src1; p.c0 = a->c0; p.c1 = a->c1; p.cin = a->c0 + a->c1;
  p.n = a->n; p.hin = a->hin; p.win = a->win;
  p.hc = a->upsample ? 2 * a->hin : a->hin;
  p.wc = a->upsample ? 2 * a->win : a->win;
  const int pad = a->ksize / 2;
  p.hout = (p.hc + 2 * pad - a->ksize) / a->stride + 1;
  p.wout = (p.wc + 2 * pad - a->ksize) / a->stride + 1;
  p.ws = static_cast<float*>(a->workspace); p.cin_pad = p.cout_pad = 0;
  p.cout = a->cout; p.dy = a->dy; p.dy_ctotal = a->dy_ctotal ? a->dy_ctotal : a->cout; p.dy_coff = a->dy_coff;
  p.ss = a->gn_scale_shift; p.silu = a->silu; p.dw = a->dw;
  p.tiles_x = p.wout / 32; p.tiles_y = p.hout / WG_SR; p.ntiles = p.tiles_x * p.tiles_y * p.n; p.ci_blocks = 1;
  const bool tile_ok = (p.wout % 32 == 0) && (p.hout % WG_SR == 0) && !a->force_direct;
  if (tile_ok) {
    const int k = a->ksize, s = a->stride, u = a->upsample;
    const bool small_ci = p.cin <= 32;
    if (k == 3 && s == 1 && u == 0) return small_ci ? launch_wgrad<3, 1, 0, 1>(p, a->workspace_bytes, st) : launch_wgrad<3, 1, 0, 2>(p, a->workspace_bytes, st);
    if (k == 3 && s == 1 && u == 1) return small_ci ? launch_wgrad<3, 1, 1, 1>(p, a->workspace_bytes, st) : launch_wgrad<3, 1, 1, 2>(p, a->workspace_bytes, st);
    if (k == 3 && s == 2) return launch_wgrad<3, 2, 0, 1>(p, a->workspace_bytes, st);
    if (k == 1 && s == 1 && u == 0) return small_ci ? launch_wgrad<1, 1, 0, 1>(p, a->workspace_bytes, st) : launch_wgrad<1, 1, 0, 2>(p, a->workspace_bytes, st);
  }
  const int64_t total = (int64_t)p.cout * p.cin * a->ksize * a->ksize;
  hipLaunchKernelGGL(conv_wgrad_direct_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, p, a->ksize,
                     a->stride, a->upsample);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_conv2d_wgrad_workspace_bytes(const dsg_conv_wgrad_args* a, size_t* bytes) {
  DSG_CHECK_ARG(a && bytes, "dsg_conv2d_wgrad_workspace_bytes: NULL argument");
  const int hc = a->upsample ? 2 * a->hin : a->hin, wc = a->upsample ? 2 * a->win : a->win;
  const int pad = a->ksize / 2;
  const int hout = (hc + 2 * pad - a->ksize) / a->stride + 1, wout = (wc + 2 * pad - a->ksize) / a->stride + 1;
  *bytes = a->force_direct ? 0 : dsg::wgrad_ws_bytes(a->c0 + a->c1, a->cout, a->ksize, a->stride, hout, wout, a->n);
  return DSG_OK;
}
