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
#include "dsg_h16.h"
#include <algorithm>
#include <type_traits>

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
  float* dysum_ws;  // optional (split kernel): [run][cout_pad] sums of dY over the run's pixels -- the bias / time-embedding
                    // gradients as a by-product of the tiles that pass through the kernel anyway
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

// The per-run sums of dY a weight-gradient kernel leaves behind (bias / time-embedding gradients) are finished by extra
// workgroups of the SAME launch that reduces its partial slabs: out[n][co] = sum over the image's runs, and bias_grad[co] +=
// their sum over the batch (fp64 across the batch, fixed order).  Was a launch of its own after every conv: 63 per training step.
struct DysumJob {
  const float* part;  // [n][slabs_per_image][cout]; nullptr = no job
  int slabs_per_image, cout, n;
  float* out;
  int out_stride;
  float* bias_grad;   // may be null
  int blocks;         // cdiv(cout, 32)
  int parts;          // 1; 4 = every slab row holds [4 pixel parities][cout] (the folded up-sampler form): summed here
};
// block b of the job: 32 channels x 8 image groups
__device__ __forceinline__ void dysum_job_block(const DysumJob& j, int b) {
  __shared__ double red[8][32];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5, co = b * 32 + cl;
  double acc = 0.0;
  if (co < j.cout) {
    for (int ni = g; ni < j.n; ni += 8) {
      float t = 0.f;
      for (int k = 0; k < j.slabs_per_image; ++k)
        for (int q = 0; q < j.parts; ++q) t += j.part[(((size_t)ni * j.slabs_per_image + k) * j.parts + q) * j.cout + co];
      j.out[(size_t)ni * j.out_stride + co] = t;
      acc += (double)t;
    }
  }
  if (j.bias_grad == nullptr) return;  // (uniform)
  red[g][cl] = acc;
  __syncthreads();
  if (g == 0 && co < j.cout) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][cl];
    j.bias_grad[co] += (float)t;
  }
}

// dw[co][ci][tp] += sum_s ws[s][tp][ci][co]; 16-byte reads (cout_pad is a multiple of 64).  A workgroup owns 64
// consecutive float4 elements; its four waves each sum every fourth slab (four loads in flight per lane), the four
// partial sums meet in LDS and are added in a fixed order -- deterministic, and the small layers (9 K elements, up to
// 256 slabs) still put four times as many waves on the chip as one thread per element would.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int nslab, int taps, int cin,
                                                           int cout, int cin_pad, int cout_pad,
                                                           float* __restrict__ dw, DysumJob job, int main_blocks) {
  if ((int)blockIdx.x >= main_blocks) {  // (the launch's extra workgroups)
    dysum_job_block(job, blockIdx.x - main_blocks);
    return;
  }
  __shared__ float4 part[4][64];
  const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t i4 = blockIdx.x * (int64_t)64 + e;
  const int64_t slab = (int64_t)taps * cin_pad * cout_pad;
  const bool in = i4 * 4 < slab;
  const int64_t i = in ? i4 * 4 : 0;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in) {
    int k = sl;
    for (; k + 12 < nslab; k += 16) {  // four of this wave's slabs at a time
      const float4 a = *reinterpret_cast<const float4*>(ws + (int64_t)k * slab + i);
      const float4 b = *reinterpret_cast<const float4*>(ws + (int64_t)(k + 4) * slab + i);
      const float4 c = *reinterpret_cast<const float4*>(ws + (int64_t)(k + 8) * slab + i);
      const float4 d = *reinterpret_cast<const float4*>(ws + (int64_t)(k + 12) * slab + i);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
      s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; k < nslab; k += 4) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)k * slab + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[sl][e] = s;
  __syncthreads();
  if (sl != 0 || !in) return;
  const float4 p0 = part[0][e], p1 = part[1][e], p2 = part[2][e], p3 = part[3][e];
  const float sv[4] = {(p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                       (p0.w + p1.w) + (p2.w + p3.w)};
  const int co = (int)(i % cout_pad);
  const int ci = (int)((i / cout_pad) % cin_pad);
  const int tp = (int)(i / ((int64_t)cout_pad * cin_pad));
  if (co >= cout || ci >= cin) return;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (co + q < cout) dw[((size_t)(co + q) * cin + ci) * taps + tp] += sv[q];
}

// The same reduction for the layers with FEW slabs and a big gradient (256 / 512 channels: 2.4 M elements, 8-32 slabs),
// where the kernel above is dominated by its output: each lane's four results go to [co][ci][tp] addresses a whole
// cin x taps row apart -- scattered 4-byte read-modify-writes.  Here a workgroup owns a 64 co x 4 ci x all-taps tile: its
// four waves each sum every fourth slab (thread = (4 co, ci), one float4 per tap: 256-byte runs along co), the partial tiles
// meet in LDS in a fixed order, and dw is updated in runs of 4 ci x taps contiguous floats per co.
constexpr int WRT_CO = 64, WRT_CI = 4;  // (16 lanes x float4 along co, 4 ci rows per wave: 256-byte runs, 16 bytes per lane)
__global__ __launch_bounds__(256) void wgrad_reduce_t_kernel(const float* __restrict__ ws, int nslab, int taps, int cin,
                                                             int cout, int cin_pad, int cout_pad,
                                                             float* __restrict__ dw, DysumJob job, int main_rows) {
  if ((int)blockIdx.y >= main_rows) {  // (the launch's extra rows of workgroups)
    const int b = (blockIdx.y - main_rows) * gridDim.x + blockIdx.x;
    if (b < job.blocks) dysum_job_block(job, b);
    return;
  }
  __shared__ float tile[4][WRT_CO][WRT_CI * 9 + 1];  // [slab wave][co][ci * taps + tp]
  const int co0 = blockIdx.x * WRT_CO, ci0 = blockIdx.y * WRT_CI;
  const int t = threadIdx.x, sl = t >> 6, co = (t & 15) * 4, ci = (t >> 4) & 3;
  const int64_t tstride = (int64_t)cin_pad * cout_pad, slab = (int64_t)taps * tstride;
  float4 acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) acc[tp] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ci0 + ci < cin_pad && co0 + co < cout_pad) {  // (cout_pad % 64 == 0: a float4 never straddles the row's end)
    const float* p = ws + (int64_t)sl * slab + (int64_t)(ci0 + ci) * cout_pad + co0 + co;
    int k = sl;
    for (; k + 4 < nslab; k += 8, p += 8 * slab) {  // two of this wave's slabs per trip: 18 loads in flight (256 workgroups of
#pragma unroll                                  // four waves are one per CU: the loop is a chain of round trips)
      for (int tp = 0; tp < 9; ++tp)
        if (tp < taps) {
          const float4 v = *reinterpret_cast<const float4*>(p + tp * tstride);
          const float4 u = *reinterpret_cast<const float4*>(p + 4 * slab + tp * tstride);
          acc[tp].x += v.x; acc[tp].y += v.y; acc[tp].z += v.z; acc[tp].w += v.w;
          acc[tp].x += u.x; acc[tp].y += u.y; acc[tp].z += u.z; acc[tp].w += u.w;
        }
    }
    for (; k < nslab; k += 4, p += 4 * slab) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
        if (tp < taps) {
          const float4 v = *reinterpret_cast<const float4*>(p + tp * tstride);
          acc[tp].x += v.x; acc[tp].y += v.y; acc[tp].z += v.z; acc[tp].w += v.w;
        }
    }
  }
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
    if (tp < taps) {
      tile[sl][co][ci * taps + tp] = acc[tp].x;
      tile[sl][co + 1][ci * taps + tp] = acc[tp].y;
      tile[sl][co + 2][ci * taps + tp] = acc[tp].z;
      tile[sl][co + 3][ci * taps + tp] = acc[tp].w;
    }
  __syncthreads();
  const int run = WRT_CI * taps;  // contiguous floats per co in dw
  for (int o = t; o < WRT_CO * run; o += 256) {
    const int c = o / run, r = o - c * run;
    if (co0 + c < cout && ci0 + r / taps < cin)
      dw[((size_t)(co0 + c) * cin + ci0) * taps + r] += (tile[0][c][r] + tile[1][c][r]) + (tile[2][c][r] + tile[3][c][r]);
  }
}

static DysumJob dysum_job(const float* part, int slabs_per_image, int cout, int n, float* out, int out_stride, float* bias_grad) {
  DysumJob j;
  j.part = part; j.slabs_per_image = slabs_per_image; j.cout = cout; j.n = n; j.out = out; j.out_stride = out_stride;
  j.bias_grad = bias_grad; j.blocks = part ? cdiv(cout, 32) : 0; j.parts = 1;
  return j;
}

static void launch_wgrad_reduce(const float* ws, int nslab, int taps, int cin, int cout, int cin_pad, int cout_pad, float* dw,
                                hipStream_t st, const DysumJob* job = nullptr) {
  const int64_t slab = (int64_t)taps * cin_pad * cout_pad;
  const DysumJob j = job ? *job : dysum_job(nullptr, 0, 0, 0, nullptr, 0, nullptr);
  if (nslab <= 32 && slab >= (int64_t)256 * 256 && cin_pad % WRT_CI == 0) {
    const int gx = cdiv(cout_pad, WRT_CO), rows = cdiv(cin_pad, WRT_CI);
    hipLaunchKernelGGL(wgrad_reduce_t_kernel, dim3(gx, rows + cdiv(j.blocks, gx)), dim3(256), 0, st, ws, nslab, taps,
                       cin, cout, cin_pad, cout_pad, dw, j, rows);
  } else {
    const int main_blocks = (int)cdiv64(slab / 4, 64);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(main_blocks + j.blocks)), dim3(256), 0, st, ws, nslab, taps, cin, cout,
                       cin_pad, cout_pad, dw, j, main_blocks);
  }
}

// Reduce pass of the folded up-sampler form (Wgrad16P.fold_co): ws[slab][tap (ty, tx)][ci][parity p][co] -> dw[co][ci][ky][kx] +=
// sum over the slabs (in slab order) of the four (tap, parity) terms that weight reads x through: parity py of an output row
// 2y + py reads x row y + floor((py - 1 + ky) / 2), which is the parity's tap ty = (py == 0 ? ky != 0 : ky == 2); the same in x.
// One thread per (ci, co): 16 coalesced reads per slab (co fastest), nine read-modify-writes of dw.
__global__ __launch_bounds__(256) void wgrad_fold_reduce_kernel(const float* __restrict__ ws, int nslab, int cin, int co,
                                                                float* __restrict__ dw, DysumJob job, int main_blocks) {
  if ((int)blockIdx.x >= main_blocks) {  // (the launch's extra workgroups)
    dysum_job_block(job, blockIdx.x - main_blocks);
    return;
  }
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= (int64_t)cin * co) return;
  const int c = (int)(i % co), ci = (int)(i / co);
  const int64_t tstride = (int64_t)cin * 4 * co, slab = 4 * tstride;
  float h[4][4];  // [tap][parity]
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) h[t][q] = 0.f;
  const float* base = ws + (int64_t)ci * 4 * co + c;
  for (int k = 0; k < nslab; ++k, base += slab) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) h[t][q] += base[t * tstride + q * co];
  }
  float* out = dw + ((size_t)c * cin + ci) * 9;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float v = 0.f;
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int ty = py == 0 ? (ky != 0) : (ky == 2), tx = px == 0 ? (kx != 0) : (kx == 2);
          v += h[ty * 2 + tx][py * 2 + px];
          // (scalar adds: left to the SLP vectoriser these sums became v_pk_add_f32 ... op_sel:[0,1], the form that returns wrong
          //  values on lanes 48-63 beside another wave's MFMAs -- tests/test_isa_policy.py)
          asm volatile("" : "+v"(v));
        }
      out[ky * 3 + kx] += v;
    }
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


// ---------------------------------------------------------------------------------------------------
// The same weight gradient on the f16 matrix cores with fp32-class accuracy ("fp16x2 split", see conv_h2.hip):
// both operands are activations here, so both are split on the way into LDS --
//   A[ci][px] = act(gn(x)) = a1 + a2 * 2^-11,   dY[co][px] = b1 + b2 * 2^-11,
//   dW += a1*b1 (acc_hi)  +  2^-11 * (a1*b2 + a2*b1) (acc_lo)        3 v_mfma_f32_32x32x16_f16 per 16 pixels
// K is the pixel index: an MFMA's k-group of 8 is 8 consecutive pixels of a row, one 16-byte LDS read per lane.
// A tap (dy, dx) reads the patch shifted by dx pixels, which would break that read's alignment, so the patch is kept
// in LDS three times, pre-shifted by dx = 0, 1, 2 (each staging thread holds 10 consecutive pixels and writes the
// three 8-pixel windows); dy only changes the row.
// Workgroup: 32 ci x 64 co x 9 taps; wave w owns co tile (w & 1) and taps 0..4 (w < 2) or 5..8: 10 / 8 accumulator
// tiles (hi + lo).  Stage = 2 output rows x 32 columns of one image (4 k-steps of 16 pixels); LDS is double-buffered
// with one barrier per stage, the next tile's values are fetched a stage ahead and converted / written between the
// MFMAs.  Row strides are padded by 16 bytes: the 16 lanes of a ds_read_b128 group fall on 16 different bank quads.
// Split-K partials go to the same workspace layout as the fp32 kernel (one slab per grid.y), then wgrad_reduce_kernel.
// ---------------------------------------------------------------------------------------------------
typedef _Float16 whalf8 __attribute__((ext_vector_type(8)));
typedef float wf32x16 __attribute__((ext_vector_type(16)));

// A lives in a ring of 6 patch rows (4 in use by the current stage + the 2 being written for the next one): a
// workgroup walks consecutive row pairs of one 32-column strip of one image, so every input row is converted once.
constexpr int WH_SLOTS = 6;
constexpr int WH_ASTR = WH_SLOTS * 32 + 8;          // halfs per (shift, piece, ci): 6 rows x 32 columns + pad
constexpr int WH_A_HALFS = 3 * 2 * 32 * WH_ASTR;    // [shift 3][piece 2][ci 32]
constexpr int WH_DSTR = 64 + 8;                     // halfs per (piece, co): 2 rows x 32 columns + pad
constexpr int WH_D_HALFS = 2 * 64 * WH_DSTR;        // [piece 2][co 64], double-buffered
constexpr int WH_LDS_BYTES = (WH_A_HALFS + 2 * WH_D_HALFS) * 2;

// Workgroup -> (pair = (ci block, co block), slab = run of pixels) for the 3x3 weight-gradient kernels, grid = (pairs, slabs).
// Workgroups go to the 8 XCDs in turn by their linear id, each XCD with an L2 of its own.  With the pair index fast the
// (ci, co) pairs of one run -- which all stream the SAME rows of x and dY -- were spread over every XCD: each L2 pulled the
// run through HBM for itself (measured: 779 MB fetched per launch for 134 MB of tensors, profiles/r03_pmc_traffic_train_bf16).
// Here XCD k takes the runs k, k + 8, ... and walks all pairs of a run on consecutive ids of its own: they run at the same
// time behind one L2 and the run's rows come from HBM once.  (A bijection of the grid whenever slabs % 8 == 0.)
// Round 4: the runs that share INPUT ROWS -- the tiles_x column strips of one (image, row split) -- go to one XCD as well, next to
// each other in time.  A strip's row is exactly one 128-byte line of an fp32 [N,C,H,W] plane, and its two halo columns live in the
// NEIGHBOURING strips' lines: with neighbouring strips on different XCDs every L2 fetched three lines per row where one is the
// strip's own (5.8 GB read for 2.6 GB of tensors on the 64-cout layers of the 256 x 256 level, which run at the memory system's
// rate: profiles/r04_pmc_traffic_train_fp32.json).  slab = (n * tiles_x + tx) * nrs + rs; group = (n, rs); XCD k takes the groups
// k, k + 8, ... and walks (group, tx, pair) with the pair fastest.  (A bijection when the number of groups is a multiple of 8;
// nrs = 0: the caller's slabs are not strips -- several strips per workgroup -- and keep the plain run mapping.)
__device__ __forceinline__ void wgrad_xcd_ids(int& pair, int& slab, int nrs = 0, int tiles_x = 0) {
  pair = blockIdx.x;
  slab = blockIdx.y;
#ifndef DSG_WGRAD_NO_XCD
  const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
  const unsigned xcd = lin & 7, j = lin >> 3;
#ifndef DSG_WGRAD_NO_STRIP_GROUPS
  if (nrs > 0 && tiles_x > 1) {
    const unsigned groups = gridDim.y / tiles_x;  // images x row splits
    if ((groups & 7) == 0 && groups * tiles_x == gridDim.y) {
      pair = (int)(j % gridDim.x);
      const unsigned tx = (j / gridDim.x) % tiles_x, gid = xcd + 8 * (j / (gridDim.x * tiles_x));
      const unsigned n = gid / nrs, rs = gid - n * nrs;
      slab = (int)((n * tiles_x + tx) * nrs + rs);
      return;
    }
  }
#endif
  if ((gridDim.y & 7) == 0) {
    slab = (int)(xcd + 8 * (j / gridDim.x));
    pair = (int)(j % gridDim.x);
  }
#endif
}

// grid = (ci blocks x co blocks, strips x row splits); p.tiles_y = stages (row pairs) per image, p.ci_blocks as usual,
// p.ntiles = row splits per strip (reused field), strips = n * tiles_x
__global__ __launch_bounds__(256, 1) void conv_wgrad_h2_kernel(WgradP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  _Float16* ab = reinterpret_cast<_Float16*>(wsm);
  _Float16* dbase = ab + WH_A_HALFS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int cot = wave & 1;
  const bool first = wave < 2;  // taps 0..4; the other pair 5..8

  int pair_id, slab_id;
  wgrad_xcd_ids(pair_id, slab_id, p.ntiles, p.tiles_x);
  const int cib = pair_id % p.ci_blocks;
  const int cob = pair_id / p.ci_blocks;
  const int ci0 = cib * 32, co0 = cob * WG_CO;
  const int plane = p.hin * p.win;
  const int oplane = p.hout * p.wout;
  const bool has_ss = p.ss != nullptr;
  const bool do_silu = has_ss && p.silu;

  // this workgroup's run: strip (image n, column tile tx), stages [s0, s1)
  const int nrs = p.ntiles;
  const int strip = slab_id / nrs, rs = slab_id - strip * nrs;
  const int n = strip / p.tiles_x, tx = strip - n * p.tiles_x;
  const int ox0 = tx * 32;
  const int per = (p.tiles_y + nrs - 1) / nrs;
  const int s0 = rs * per, s1 = min(p.tiles_y, s0 + per);

  // buffer descriptors (range-checked: reads before / past the tensor return 0, nothing faults) -- the ci block sits
  // entirely in one of the two concatenated sources (c0 % 32 == 0)
  const bool in0 = ci0 < p.c0;
  const float* srcb = in0 ? p.src0 + ((size_t)n * p.c0 + ci0) * plane : p.src1 + ((size_t)n * p.c1 + (ci0 - p.c0)) * plane;
  const float* src_all = in0 ? p.src0 : p.src1;
  const size_t src_bytes = (size_t)p.n * (in0 ? p.c0 : p.c1) * plane * 4;
  const int src_off = (int)((srcb - src_all) * 4);
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src_all), 0, (int)src_bytes, 0x00020000);
  const float* dyb = p.dy + ((size_t)n * p.dy_ctotal + p.dy_coff + co0) * oplane;
  const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dyb), 0, WG_CO * oplane * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t s_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_ss ? p.ss + ((size_t)n * p.cin + ci0) * 2 : p.dy), 0, has_ss ? 32 * 8 : 0, 0x00020000);

  // staging items.  A: (ci, row of the pair, 8-column octet) -- one per thread; dY: (co, octet of the 2x32 stage) x 2
  const int a_oct = tid & 3, a_rr = (tid >> 2) & 1, a_ci = tid >> 3;
  const int a_voff = src_off + (a_ci * plane + a_rr * p.win + ox0 + a_oct * 8 - 1) * 4;  // + (2k-1)*win*4 per row pair
  const unsigned a_colmask = 0x3FFu & ~((ox0 == 0 && a_oct == 0) ? 1u : 0u) & ~((ox0 + 32 == p.wc && a_oct == 3) ? 0x200u : 0u);
  float sca = 1.f, sha = 0.f;
  if (has_ss) {
    const float2 s2 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(s_rs, a_ci * 8, 0, 0));
    sca = s2.x;
    sha = s2.y;
  }
  int d_voff[2], d_lds[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int it = tid + 256 * u;
    const int oct = it & 7, co = it >> 3;
    d_voff[u] = (co * oplane + (oct >> 2) * p.wout + ox0 + (oct & 3) * 8) * 4;  // + 2s*wout*4 per stage
    d_lds[u] = co * WH_DSTR + oct * 8;
  }

  float xa[10];
  float4 xd[2][2];
  unsigned va = 0;
  auto load_rows = [&](int k) {  // input rows 2k-1, 2k of the strip
    const int off = a_voff + (2 * k - 1) * p.win * 4;
    // columns 0..7 of the octet as two aligned 16-byte loads, its left / right neighbours as single dwords (a
    // 16-byte access that starts before the tensor would be dropped as a whole by the range check)
    const float4 q0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, off + 4, 0, 0));
    const float4 q1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, off + 20, 0, 0));
    xa[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rs, off, 0, 0));
    xa[9] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rs, off + 36, 0, 0));
    xa[1] = q0.x; xa[2] = q0.y; xa[3] = q0.z; xa[4] = q0.w;
    xa[5] = q1.x; xa[6] = q1.y; xa[7] = q1.z; xa[8] = q1.w;
    va = ((unsigned)(2 * k - 1 + a_rr) < (unsigned)p.hc) ? a_colmask : 0u;
  };
  auto load_dy = [&](int s) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int off = d_voff[u] + 2 * s * p.wout * 4;
      xd[u][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, off, 0, 0));
      xd[u][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, off + 16, 0, 0));
    }
  };
  auto split2 = [](float v0, float v1, unsigned& hi, unsigned& lo) {  // two values -> packed (hi, hi), (lo, lo)
    const _Float16 a0 = (_Float16)v0, a1 = (_Float16)v1;
    const _Float16 b0 = (_Float16)((v0 - (float)a0) * 2048.0f), b1 = (_Float16)((v1 - (float)a1) * 2048.0f);
    hi = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, a1) << 16);
    lo = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
  };
  // the values of load_rows(k) -> ring slots (2k) % 6, (2k + 1) % 6, three shifts; in two parts so that the work can
  // be dealt over two k-steps of MFMAs (part 0: element pairs 0..2, part 1: pairs 3..4, the shifts and the writes)
  unsigned ph[5], pl[5];
  auto commit_rows = [&](int k, int part) {
#pragma unroll
    for (int j2 = (part == 1 ? 3 : 0); j2 < (part == 0 ? 3 : 5); ++j2) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * j2 + e;
        float x = xa[j] * sca + sha;
        const float sx = silu_fast_b(x);
        x = do_silu ? sx : x;
        v[e] = ((va >> j) & 1u) ? x : 0.f;
      }
      split2(v[0], v[1], ph[j2], pl[j2]);
    }
    if (part == 0) return;
    const int slot = (2 * k) % WH_SLOTS + a_rr;  // (2k % 6 is even, so + rr stays inside the ring)
    _Float16* dst = ab + a_ci * WH_ASTR + slot * 32 + a_oct * 8;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      uint4 wh, wl;
      if (s == 1) {
        wh = make_uint4(__builtin_amdgcn_alignbit(ph[1], ph[0], 16), __builtin_amdgcn_alignbit(ph[2], ph[1], 16),
                        __builtin_amdgcn_alignbit(ph[3], ph[2], 16), __builtin_amdgcn_alignbit(ph[4], ph[3], 16));
        wl = make_uint4(__builtin_amdgcn_alignbit(pl[1], pl[0], 16), __builtin_amdgcn_alignbit(pl[2], pl[1], 16),
                        __builtin_amdgcn_alignbit(pl[3], pl[2], 16), __builtin_amdgcn_alignbit(pl[4], pl[3], 16));
      } else {
        const int o = s >> 1;
        wh = make_uint4(ph[o], ph[o + 1], ph[o + 2], ph[o + 3]);
        wl = make_uint4(pl[o], pl[o + 1], pl[o + 2], pl[o + 3]);
      }
      *reinterpret_cast<uint4*>(dst + (s * 2 + 0) * 32 * WH_ASTR) = wh;
      *reinterpret_cast<uint4*>(dst + (s * 2 + 1) * 32 * WH_ASTR) = wl;
    }
  };
  float dsum[2] = {0.f, 0.f};  // this thread's two (co, pixel octet) items of dY, summed over the run (cnt: 1 inside it, 0 past it)
  auto commit_dy = [&](int par, int u0, int u1, float cnt) {
    _Float16* db = dbase + par * WH_D_HALFS;
#pragma unroll
    for (int u = u0; u < u1; ++u) {
      dsum[u] += cnt * (((xd[u][0].x + xd[u][0].y) + (xd[u][0].z + xd[u][0].w)) +
                        ((xd[u][1].x + xd[u][1].y) + (xd[u][1].z + xd[u][1].w)));
      uint4 dh, dl;
      split2(xd[u][0].x, xd[u][0].y, dh.x, dl.x);
      split2(xd[u][0].z, xd[u][0].w, dh.y, dl.y);
      split2(xd[u][1].x, xd[u][1].y, dh.z, dl.z);
      split2(xd[u][1].z, xd[u][1].w, dh.w, dl.w);
      *reinterpret_cast<uint4*>(db + d_lds[u]) = dh;
      *reinterpret_cast<uint4*>(db + 64 * WH_DSTR + d_lds[u]) = dl;
    }
  };

  wf32x16 acc_hi[5], acc_lo[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc_hi[t][r] = 0.f;
      acc_lo[t][r] = 0.f;
    }

  // prologue: row pairs s0 and s0+1 and dY(s0) into LDS; row pair s0+2 and dY(s0+1) into registers
  if (s0 < s1) {
    load_rows(s0);
    load_dy(s0);
    commit_rows(s0, 2);
    load_rows(s0 + 1);
    commit_dy(0, 0, 2, 1.f);
    commit_rows(s0 + 1, 2);
    load_rows(s0 + 2);               // (past the image: masked to zero)
    if (s0 + 1 < s1) load_dy(s0 + 1);
  }
  __syncthreads();

  const _Float16* a_lane = ab + l31 * WH_ASTR + half * 8;
  const _Float16* d_lane = dbase + (cot * 32 + l31) * WH_DSTR + half * 8;
  // (one loop per tap half, not one loop with a branch inside: with the branch inside, the accumulators of the two
  // variants met in phi nodes and were copied between VGPRs and AGPRs every stage)
  auto run = [&](auto first_tag) {
  for (int s = s0; s < s1; ++s) {
    const int par = (s - s0) & 1;
    const bool more = s + 1 < s1;
    // ring slots of the 4 input rows of this stage: row j (0..3) = input row 2s - 1 + j -> slot (2s + j) % 6
    int slot_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) slot_off[j] = ((2 * s + j) % WH_SLOTS) * 32;
    const _Float16* dl = d_lane + par * WH_D_HALFS;
    {
      constexpr bool FIRST = decltype(first_tag)::value;
      constexpr int T0 = FIRST ? 0 : 5, NTP = FIRST ? 5 : 4;
      // Operands are fetched one k-step ahead into the other half of a register double buffer: the reads of k-step
      // kk+1 precede k-step kk's staging writes in program order, so kk's MFMAs depend on registers only and the
      // scheduler may interleave them with the staging work (LDS reads after a possibly-aliasing write cannot move).
      whalf8 fa[2][NTP][2], fb[2][2];
      auto frags = [&](int kk, int fp) {
        const int orow = kk >> 1, colg = (kk & 1) * 16;
        fb[fp][0] = *reinterpret_cast<const whalf8*>(dl + orow * 32 + colg);
        fb[fp][1] = *reinterpret_cast<const whalf8*>(dl + 64 * WH_DSTR + orow * 32 + colg);
#pragma unroll
        for (int tp = 0; tp < NTP; ++tp) {
          const int dy = (T0 + tp) / 3, dx = (T0 + tp) % 3;
          const _Float16* ap = a_lane + slot_off[orow + dy] + colg;
          fa[fp][tp][0] = *reinterpret_cast<const whalf8*>(ap + (dx * 2 + 0) * 32 * WH_ASTR);
          fa[fp][tp][1] = *reinterpret_cast<const whalf8*>(ap + (dx * 2 + 1) * 32 * WH_ASTR);
        }
      };
      frags(0, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        __builtin_amdgcn_sched_barrier(0);
        if (kk < 3) frags(kk + 1, (kk + 1) & 1);
        // next stage's staging is dealt over the four k-steps and pinned between their MFMAs: patch rows (two
        // parts), then the two dY items; each fetch of the stage after it follows its commit at once
        if (kk == 0) commit_rows(s + 2, 0);
        if (kk == 1) {
          commit_rows(s + 2, 1);
          load_rows(s + 3);
        }
        // (unconditional: past the run's last stage the values are never read and the loads are range-checked to
        // zero -- a branch here would fence the scheduler)
        if (kk == 2) commit_dy(par ^ 1, 0, 1, more ? 1.f : 0.f);
        if (kk == 3) {
          commit_dy(par ^ 1, 1, 2, more ? 1.f : 0.f);
          load_dy(s + 2);
        }
        const int fp = kk & 1;
#pragma unroll
        for (int tp = 0; tp < NTP; ++tp) {
          acc_hi[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fp][tp][0], fb[fp][0], acc_hi[tp], 0, 0, 0);
          acc_lo[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fp][tp][0], fb[fp][1], acc_lo[tp], 0, 0, 0);
          acc_lo[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fp][tp][1], fb[fp][0], acc_lo[tp], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 3 * NTP; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  };
  if (first) run(std::true_type{});
  else run(std::false_type{});

  if (p.dysum_ws != nullptr && cib == 0) {  // one ci block per co block owns the by-product; 8 neighbouring lanes share a co
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float t = dsum[u];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      if ((tid & 7) == 0) p.dysum_ws[(size_t)slab_id * p.cout_pad + co0 + ((tid + 256 * u) >> 3)] = t;
    }
  }
  // epilogue: D[ci][co = l31]; partials to this run's slab [tap][ci][co]
  const int co = co0 + cot * 32 + l31;
  float* wsb = p.ws + (size_t)slab_id * 9 * p.cin_pad * p.cout_pad;
  const int t0 = first ? 0 : 5, ntp = first ? 5 : 4;
#pragma unroll
  for (int tp = 0; tp < 5; ++tp) {
    if (tp < ntp) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        wsb[((size_t)(t0 + tp) * p.cin_pad + ci) * p.cout_pad + co] = acc_hi[tp][r] + acc_lo[tp][r] * (1.0f / 2048.0f);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The same kernel with a 32 ci x 128 co workgroup (cout % 128 == 0; round 4, dsg_set_tuning key 31).  The 32 x 64 workgroup
// re-reads, re-activates and re-splits A once per 64 output channels and its waves see 12 fragment reads per 15 (12) matrix
// instructions, waves 0-1 carrying five taps against the others' four.  Here wave w = (co pair cp = w & 1, tap group g = w >> 1)
// owns the co tiles 2cp, 2cp + 1 with the taps 0..3 (g = 0) or 5..8 (g = 1), and the centre tap for co tile 2cp + g: nine
// (tap, co tile) units on every wave -- 27 matrix instructions per k-step on every SIMD, fed by 14 fragment reads (an A fragment
// serves two co tiles), the activation arithmetic of a stage spread over 108 matrix instructions instead of 60 (48), x read
// cout / 128 times.  18 accumulator tiles (hi + lo) are 288 registers: sixteen live in the 256 AGPRs behind the builtin, the
// centre unit's two are pinned to architectural registers through the instruction as inline asm (conv_wgrad16_kernel's device).
// Per (ci, co, tap) the products are accumulated over a run's pixels in the 32 x 64 kernel's order: equal runs, equal bits.
// LDS: the A ring as before (76.8 KB) + dY [buffer 2][piece 2][co 128][2 rows x 32 columns] (73.7 KB).
// ---------------------------------------------------------------------------------------------------
constexpr int WW_CO = 128;
constexpr int WW_D_HALFS = 2 * WW_CO * WH_DSTR;
constexpr int WW_LDS_BYTES = (WH_A_HALFS + 2 * WW_D_HALFS) * 2;

__device__ __forceinline__ void mma_f16_vacc(const whalf8& a, const whalf8& b, wf32x16& c) {
  asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

__global__ __launch_bounds__(256, 1) void conv_wgrad_h2w_kernel(WgradP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  _Float16* ab = reinterpret_cast<_Float16*>(wsm);
  _Float16* dbase = ab + WH_A_HALFS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int cp = wave & 1;
  const bool first = wave < 2;  // tap group 0: taps 0..3 and the centre tap for co tile 2cp; group 1: 5..8, centre for 2cp + 1

  int pair_id, slab_id;
  wgrad_xcd_ids(pair_id, slab_id, p.ntiles, p.tiles_x);
  const int cib = pair_id % p.ci_blocks;
  const int cob = pair_id / p.ci_blocks;
  const int ci0 = cib * 32, co0 = cob * WW_CO;
  const int plane = p.hin * p.win;
  const int oplane = p.hout * p.wout;
  const bool has_ss = p.ss != nullptr;
  const bool do_silu = has_ss && p.silu;

  const int nrs = p.ntiles;
  const int strip = slab_id / nrs, rs = slab_id - strip * nrs;
  const int n = strip / p.tiles_x, tx = strip - n * p.tiles_x;
  const int ox0 = tx * 32;
  const int per = (p.tiles_y + nrs - 1) / nrs;
  const int s0 = rs * per, s1 = min(p.tiles_y, s0 + per);

  const bool in0 = ci0 < p.c0;
  const float* srcb = in0 ? p.src0 + ((size_t)n * p.c0 + ci0) * plane : p.src1 + ((size_t)n * p.c1 + (ci0 - p.c0)) * plane;
  const float* src_all = in0 ? p.src0 : p.src1;
  const size_t src_bytes = (size_t)p.n * (in0 ? p.c0 : p.c1) * plane * 4;
  const int src_off = (int)((srcb - src_all) * 4);
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src_all), 0, (int)src_bytes, 0x00020000);
  const float* dyb = p.dy + ((size_t)n * p.dy_ctotal + p.dy_coff + co0) * oplane;
  const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dyb), 0, WW_CO * oplane * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t s_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_ss ? p.ss + ((size_t)n * p.cin + ci0) * 2 : p.dy), 0, has_ss ? 32 * 8 : 0, 0x00020000);

  // staging items.  A: (ci, row of the pair, 8-column octet) -- one per thread; dY: (co, octet of the 2x32 stage) x 4
  const int a_oct = tid & 3, a_rr = (tid >> 2) & 1, a_ci = tid >> 3;
  const int a_voff = src_off + (a_ci * plane + a_rr * p.win + ox0 + a_oct * 8 - 1) * 4;
  const unsigned a_colmask = 0x3FFu & ~((ox0 == 0 && a_oct == 0) ? 1u : 0u) & ~((ox0 + 32 == p.wc && a_oct == 3) ? 0x200u : 0u);
  float sca = 1.f, sha = 0.f;
  if (has_ss) {
    const float2 s2 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(s_rs, a_ci * 8, 0, 0));
    sca = s2.x;
    sha = s2.y;
  }
  // dY item u of this thread: co = (tid >> 3) + 32 u, octet tid & 7 (the same octet for its four items)
  const int d_oct = tid & 7;
  const int d_voff0 = ((tid >> 3) * oplane + (d_oct >> 2) * p.wout + ox0 + (d_oct & 3) * 8) * 4;  // + u * 32 * oplane * 4, + 2s*wout*4
  const int d_lds0 = (tid >> 3) * WH_DSTR + d_oct * 8;                                           // + u * 32 * WH_DSTR
  const int d_ustep = 32 * oplane * 4;

  float xa[10];
  float4 xd[4][2];
  unsigned va = 0;
  auto load_rows = [&](int k) {  // input rows 2k-1, 2k of the strip
    const int off = a_voff + (2 * k - 1) * p.win * 4;
    const float4 q0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, off + 4, 0, 0));
    const float4 q1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, off + 20, 0, 0));
    xa[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rs, off, 0, 0));
    xa[9] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rs, off + 36, 0, 0));
    xa[1] = q0.x; xa[2] = q0.y; xa[3] = q0.z; xa[4] = q0.w;
    xa[5] = q1.x; xa[6] = q1.y; xa[7] = q1.z; xa[8] = q1.w;
    va = ((unsigned)(2 * k - 1 + a_rr) < (unsigned)p.hc) ? a_colmask : 0u;
  };
  auto load_dy = [&](int s, int u0, int u1) {
#pragma unroll
    for (int u = u0; u < u1; ++u) {
      const int off = d_voff0 + u * d_ustep + 2 * s * p.wout * 4;
      xd[u][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, off, 0, 0));
      xd[u][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, off + 16, 0, 0));
    }
  };
  auto split2 = [](float v0, float v1, unsigned& hi, unsigned& lo) {
    const _Float16 a0 = (_Float16)v0, a1 = (_Float16)v1;
    const _Float16 b0 = (_Float16)((v0 - (float)a0) * 2048.0f), b1 = (_Float16)((v1 - (float)a1) * 2048.0f);
    hi = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, a1) << 16);
    lo = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
  };
  unsigned ph[5], pl[5];
  auto commit_rows = [&](int k, int part) {
#pragma unroll
    for (int j2 = (part == 1 ? 3 : 0); j2 < (part == 0 ? 3 : 5); ++j2) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * j2 + e;
        float x = xa[j] * sca + sha;
        const float sx = silu_fast_b(x);
        x = do_silu ? sx : x;
        v[e] = ((va >> j) & 1u) ? x : 0.f;
      }
      split2(v[0], v[1], ph[j2], pl[j2]);
    }
    if (part == 0) return;
    const int slot = (2 * k) % WH_SLOTS + a_rr;
    _Float16* dst = ab + a_ci * WH_ASTR + slot * 32 + a_oct * 8;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      uint4 wh, wl;
      if (s == 1) {
        wh = make_uint4(__builtin_amdgcn_alignbit(ph[1], ph[0], 16), __builtin_amdgcn_alignbit(ph[2], ph[1], 16),
                        __builtin_amdgcn_alignbit(ph[3], ph[2], 16), __builtin_amdgcn_alignbit(ph[4], ph[3], 16));
        wl = make_uint4(__builtin_amdgcn_alignbit(pl[1], pl[0], 16), __builtin_amdgcn_alignbit(pl[2], pl[1], 16),
                        __builtin_amdgcn_alignbit(pl[3], pl[2], 16), __builtin_amdgcn_alignbit(pl[4], pl[3], 16));
      } else {
        const int o = s >> 1;
        wh = make_uint4(ph[o], ph[o + 1], ph[o + 2], ph[o + 3]);
        wl = make_uint4(pl[o], pl[o + 1], pl[o + 2], pl[o + 3]);
      }
      *reinterpret_cast<uint4*>(dst + (s * 2 + 0) * 32 * WH_ASTR) = wh;
      *reinterpret_cast<uint4*>(dst + (s * 2 + 1) * 32 * WH_ASTR) = wl;
    }
  };
  float dsum[4] = {0.f, 0.f, 0.f, 0.f};
  auto commit_dy = [&](int par, int u0, int u1, float cnt) {
    _Float16* db = dbase + par * WW_D_HALFS + d_lds0;
#pragma unroll
    for (int u = u0; u < u1; ++u) {
      dsum[u] += cnt * (((xd[u][0].x + xd[u][0].y) + (xd[u][0].z + xd[u][0].w)) +
                        ((xd[u][1].x + xd[u][1].y) + (xd[u][1].z + xd[u][1].w)));
      uint4 dh, dl;
      split2(xd[u][0].x, xd[u][0].y, dh.x, dl.x);
      split2(xd[u][0].z, xd[u][0].w, dh.y, dl.y);
      split2(xd[u][1].x, xd[u][1].y, dh.z, dl.z);
      split2(xd[u][1].z, xd[u][1].w, dh.w, dl.w);
      *reinterpret_cast<uint4*>(db + u * 32 * WH_DSTR) = dh;
      *reinterpret_cast<uint4*>(db + WW_CO * WH_DSTR + u * 32 * WH_DSTR) = dl;
    }
  };

  wf32x16 acc_hi[8], acc_lo[8], cen_hi, cen_lo;  // [tap of the group 4][co tile of the pair 2]; the centre unit (pinned)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    cen_hi[r] = 0.f;
    cen_lo[r] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc_hi[t][r] = 0.f;
      acc_lo[t][r] = 0.f;
    }

  if (s0 < s1) {
    load_rows(s0);
    load_dy(s0, 0, 4);
    commit_rows(s0, 2);
    load_rows(s0 + 1);
    commit_dy(0, 0, 4, 1.f);
    commit_rows(s0 + 1, 2);
    load_rows(s0 + 2);
    if (s0 + 1 < s1) load_dy(s0 + 1, 0, 4);
  }
  __syncthreads();

  const _Float16* a_lane = ab + l31 * WH_ASTR + half * 8;
  const _Float16* d_lane = dbase + (cp * 64 + l31) * WH_DSTR + half * 8;
  auto run = [&](auto first_tag) {
  for (int s = s0; s < s1; ++s) {
    const int par = (s - s0) & 1;
    const bool more = s + 1 < s1;
    int slot_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) slot_off[j] = ((2 * s + j) % WH_SLOTS) * 32;
    const _Float16* dl = d_lane + par * WW_D_HALFS;
    {
      constexpr bool FIRST = decltype(first_tag)::value;
      constexpr int T0 = FIRST ? 0 : 5, CJ = FIRST ? 0 : 1;
      whalf8 fa[2][5][2], fb[2][2][2];
      auto frags = [&](int kk, int fp) {
        const int orow = kk >> 1, colg = (kk & 1) * 16;
#pragma unroll
        for (int cj = 0; cj < 2; ++cj) {
          fb[fp][cj][0] = *reinterpret_cast<const whalf8*>(dl + cj * 32 * WH_DSTR + orow * 32 + colg);
          fb[fp][cj][1] = *reinterpret_cast<const whalf8*>(dl + (WW_CO + cj * 32) * WH_DSTR + orow * 32 + colg);
        }
#pragma unroll
        for (int tp = 0; tp < 5; ++tp) {
          const int tap = tp < 4 ? T0 + tp : 4;
          const int dy = tap / 3, dx = tap % 3;
          const _Float16* ap = a_lane + slot_off[orow + dy] + colg;
          fa[fp][tp][0] = *reinterpret_cast<const whalf8*>(ap + (dx * 2 + 0) * 32 * WH_ASTR);
          fa[fp][tp][1] = *reinterpret_cast<const whalf8*>(ap + (dx * 2 + 1) * 32 * WH_ASTR);
        }
      };
      frags(0, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        __builtin_amdgcn_sched_barrier(0);
        if (kk < 3) frags(kk + 1, (kk + 1) & 1);
        if (kk == 0) commit_rows(s + 2, 0);
        if (kk == 1) {
          commit_rows(s + 2, 1);
          load_rows(s + 3);
        }
        if (kk == 2) {
          commit_dy(par ^ 1, 0, 2, more ? 1.f : 0.f);
          load_dy(s + 2, 0, 2);
        }
        if (kk == 3) {
          commit_dy(par ^ 1, 2, 4, more ? 1.f : 0.f);
          load_dy(s + 2, 2, 4);
        }
        const int fp = kk & 1;
        // the centre unit first (inline asm on a pinned tile: a whole k-step of other instructions follows before the
        // tile is touched again), its two low-order products behind taps 0 and 1
        mma_f16_vacc(fa[fp][4][0], fb[fp][CJ][0], cen_hi);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
          for (int cj = 0; cj < 2; ++cj) {
            acc_hi[tp * 2 + cj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fp][tp][0], fb[fp][cj][0], acc_hi[tp * 2 + cj], 0, 0, 0);
            acc_lo[tp * 2 + cj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fp][tp][0], fb[fp][cj][1], acc_lo[tp * 2 + cj], 0, 0, 0);
            acc_lo[tp * 2 + cj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fp][tp][1], fb[fp][cj][0], acc_lo[tp * 2 + cj], 0, 0, 0);
          }
          if (tp == 0) mma_f16_vacc(fa[fp][4][0], fb[fp][CJ][1], cen_lo);
          if (tp == 1) mma_f16_vacc(fa[fp][4][1], fb[fp][CJ][0], cen_lo);
        }
#pragma unroll
        for (int m = 0; m < 24; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  };
  if (first) run(std::true_type{});
  else run(std::false_type{});

  if (p.dysum_ws != nullptr && cib == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float t = dsum[u];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      if ((tid & 7) == 0) p.dysum_ws[(size_t)slab_id * p.cout_pad + co0 + (tid >> 3) + 32 * u] = t;
    }
  }
  // epilogue: D[ci][co = l31] per (tap, co tile); partials to this run's slab [tap][ci][co]
  float* wsb = p.ws + (size_t)slab_id * 9 * p.cin_pad * p.cout_pad;
  const int t0 = first ? 0 : 5;
#pragma unroll
  for (int tp = 0; tp < 4; ++tp)
#pragma unroll
    for (int cj = 0; cj < 2; ++cj) {
      const int co = co0 + (2 * cp + cj) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        wsb[((size_t)(t0 + tp) * p.cin_pad + ci) * p.cout_pad + co] = acc_hi[tp * 2 + cj][r] + acc_lo[tp * 2 + cj][r] * (1.0f / 2048.0f);
      }
    }
  {
    const int co = co0 + (2 * cp + (first ? 0 : 1)) * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      wsb[((size_t)4 * p.cin_pad + ci) * p.cout_pad + co] = cen_hi[r] + cen_lo[r] * (1.0f / 2048.0f);
    }
  }
}

static int g_wgrad_h2_wide = 1;  // dsg_set_tuning key 31 (tests / A-B runs): 0 = the 32 x 64 workgroup everywhere
void wgrad_h2_set_wide(int v) { g_wgrad_h2_wide = v; }
static bool wgrad_h2_wide(int cout) { return g_wgrad_h2_wide && cout % WW_CO == 0; }

static int g_wgrad_h2 = 1;
void wgrad_h2_set_enabled(int on) { g_wgrad_h2 = on; }

static bool wgrad_h2_eligible(const WgradP& p, int ks, int stride, int ups) {
  return g_wgrad_h2 && ks == 3 && stride == 1 && ups == 0 && p.cin % 32 == 0 && p.cout % 64 == 0 && p.wout % 32 == 0 &&
         p.hout % 2 == 0 && (p.c1 == 0 || p.c0 % 32 == 0);
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
  (void)slab;
  launch_wgrad_reduce(p.ws, nslab, G::TAPS, p.cin, p.cout, p.cin_pad, p.cout_pad, p.dw, st);
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

// runs of the fp16x2-split kernel: every (image, 32-column strip) is cut into `rsplit` runs of consecutive row pairs
static void wgrad_h2_runs(int cin, int cout, int n, int hout, int wout, int* strips, int* rsplit) {
  const int pairs = (cin / 32) * (cout / (wgrad_h2_wide(cout) ? WW_CO : WG_CO));
  *strips = n * (wout / 32);
  const int stages = hout / 2;
  const int want = std::max(1, cdiv(512, pairs));             // workgroups wanted per (ci, co) pair (2 per CU in all)
  *rsplit = std::max(1, std::min(stages, cdiv(want, *strips)));
}

static int launch_wgrad_h2(WgradP p, size_t ws_bytes, hipStream_t st, float* dy_sums = nullptr, int dy_sums_stride = 0,
                           float* dy_bias_grad = nullptr) {
  p.ci_blocks = p.cin / 32;
  const bool wide = wgrad_h2_wide(p.cout);
  const int co_blocks = p.cout / (wide ? WW_CO : WG_CO);
  const int pairs = p.ci_blocks * co_blocks;
  int strips, rsplit;
  wgrad_h2_runs(p.cin, p.cout, p.n, p.hout, p.wout, &strips, &rsplit);
  const int nslab = strips * rsplit;
  p.cin_pad = p.cin;
  p.cout_pad = p.cout;
  p.ntiles = rsplit;  // (field reused: row splits per strip)
  const size_t need = (size_t)nslab * (9 * (size_t)p.cin_pad * p.cout_pad + p.cout_pad) * sizeof(float);
  if (p.ws == nullptr || ws_bytes < need)
    return fail(DSG_ERR_WORKSPACE_TOO_SMALL, "dsg_conv2d_wgrad: workspace %zu bytes < required %zu", ws_bytes, need);
  p.dysum_ws = dy_sums ? p.ws + (size_t)nslab * 9 * p.cin_pad * p.cout_pad : nullptr;
  static bool raised = false;
  if (!raised) {
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_h2_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_h2w_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised = true;
  }
  int pi = -1;
  if (prof_on())
    pi = prof_begin(9, 2.0 * p.n * p.hout * p.wout * (double)p.cout * p.cin * 9,
                    4.0 * ((double)p.n * p.cin * p.hin * p.win + (double)p.n * p.cout * p.hout * p.wout), st);
  if (wide) hipLaunchKernelGGL(conv_wgrad_h2w_kernel, dim3(pairs, nslab), dim3(256), (size_t)WW_LDS_BYTES, st, p);
  else hipLaunchKernelGGL(conv_wgrad_h2_kernel, dim3(pairs, nslab), dim3(256), (size_t)WH_LDS_BYTES, st, p);
  DSG_LAUNCH_CHECK();
  const int64_t slab = (int64_t)9 * p.cin_pad * p.cout_pad;
  (void)slab;
  // (run index = (image, column tile, row split): an image's runs are consecutive)
  const DysumJob job = dysum_job(dy_sums ? p.dysum_ws : nullptr, p.tiles_x * rsplit, p.cout, p.n, dy_sums,
                                 dy_sums_stride ? dy_sums_stride : p.cout, dy_bias_grad);
  launch_wgrad_reduce(p.ws, nslab, 9, p.cin, p.cout, p.cin_pad, p.cout_pad, p.dw, st, &job);
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

// ---------------------------------------------------------------------------------------------------
// Pointwise (1x1) weight gradient on the same fp16x2 split: dW[co][ci] = sum over (n, pixel) of dY[co][px] * A[ci][px].
// Both operands are pixel-contiguous in [N, C, H, W], so a k-group of 8 is 8 consecutive pixels on either side and no
// shifted copies exist: a stage is 64 pixels of one image, TM ci x TN co per workgroup (128 x 128: wave w owns the
// 64 x 64 quarter (w >> 1, w & 1), four accumulator tile pairs; 64 x 64 for the channel counts 128 does not divide).
// With one tap the staging (affine, split, LDS write: ~7 VALU per element) weighs as much as the MFMAs, so the tile is
// as large as the register file allows -- every A row is staged cout / TN times, every dY row cin / TM times.  The
// (n, pixel) sequence is cut into gridDim.y runs; partial slabs [ci][co] and the fixed-order reduce as above.
// No 1x1 conv of the network has SiLU on its input (shortcuts and to_out read raw tensors, qkv the normalised one);
// the flag is honoured all the same.
// ---------------------------------------------------------------------------------------------------
constexpr int WP_STR = 64 + 8;  // halfs per (piece, channel): 64 pixels + 16 bytes of padding

__device__ __forceinline__ void wsplit2(float v0, float v1, unsigned& hi, unsigned& lo) {
  const _Float16 a0 = (_Float16)v0, a1 = (_Float16)v1;
  const _Float16 b0 = (_Float16)((v0 - (float)a0) * 2048.0f), b1 = (_Float16)((v1 - (float)a1) * 2048.0f);
  hi = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, a1) << 16);
  lo = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
}

// grid = (runs, ci blocks x co blocks); p.ci_blocks = cin / TM.  The run index is the fast one: consecutive workgroups
// go to different XCDs, so each XCD's L2 holds ITS pixel runs of A and dY once and serves every (ci, co) tile pair of
// them (with the pair index fast, every XCD pulled all of dY through its own L2)
template <int MI, int NJ>
__global__ __launch_bounds__(256, 1) void conv_wgrad_h2_pw_kernel(WgradP p) {
  constexpr int TM = 64 * MI, TN = 64 * NJ;
  constexpr int A_HALFS = 2 * TM * WP_STR, D_HALFS = 2 * TN * WP_STR, BUF_HALFS = A_HALFS + D_HALFS;  // [piece][channel][72]
  // staging items per thread: (channel row + 32 j; pixels 4 q .. 4 q + 3 and 32 + 4 q .. of the stage, q = tid & 7): a
  // 16-byte load instruction covers whole 128-byte lines
  constexpr int NA = TM / 32, ND = TN / 32, NI = NA + ND;
  static_assert(NI % 4 == 0, "items are dealt over the four k-steps of a stage");
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  _Float16* lds = reinterpret_cast<_Float16*>(wsm);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wci = wave >> 1, wco = wave & 1;
  const int cib = blockIdx.y % p.ci_blocks, cob = blockIdx.y / p.ci_blocks;
  const int ci0 = cib * TM, co0 = cob * TN;
  const int plane = p.hin * p.win;
  const int spi = plane >> 6;  // stages per image
  const int total = p.n * spi;
  const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
  const int s0 = blockIdx.x * per, s1 = min(total, s0 + per);
  const bool has_ss = p.ss != nullptr;
  const bool do_silu = has_ss && p.silu;
  const int row = tid >> 3, oct = tid & 7;

  const float* abase[NA];
  size_t astride[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int ci = ci0 + row + 32 * j;
    const bool in0 = ci < p.c0;
    abase[j] = (in0 ? p.src0 + (size_t)ci * plane : p.src1 + (size_t)(ci - p.c0) * plane) + oct * 4;
    astride[j] = (size_t)(in0 ? p.c0 : p.c1) * plane;
  }
  const float* dbase = p.dy + ((size_t)p.dy_coff + co0 + row) * plane + oct * 4;
  const size_t dstride = (size_t)p.dy_ctotal * plane;

  float4 xa[NA][2], xd[ND][2];
  float sca[NA], sha[NA];
  int ln = 0, lq = 0, lst = 0;  // (image, stage within it, stage index) of the loads issued next; never past the run
  auto load_a = [&](int j) {
    const float* q = abase[j] + (size_t)ln * astride[j] + (lq << 6);
    xa[j][0] = *reinterpret_cast<const float4*>(q);
    xa[j][1] = *reinterpret_cast<const float4*>(q + 32);
    if (has_ss) {
      const float2 s2 = *reinterpret_cast<const float2*>(p.ss + ((size_t)ln * p.cin + ci0 + row + 32 * j) * 2);
      sca[j] = s2.x;
      sha[j] = s2.y;
    }
  };
  auto load_d = [&](int j) {
    const float* q = dbase + (size_t)ln * dstride + (size_t)(32 * j) * plane + (lq << 6);
    xd[j][0] = *reinterpret_cast<const float4*>(q);
    xd[j][1] = *reinterpret_cast<const float4*>(q + 32);
  };
  auto advance = [&]() {
    if (lst + 1 < s1) {
      ++lst;
      if (++lq == spi) {
        lq = 0;
        ++ln;
      }
    }
  };
  auto act = [&](float x, int j) {
    if (has_ss) {
      x = x * sca[j] + sha[j];
      if (do_silu) x = silu_fast_b(x);
    }
    return x;
  };
  auto commit_a = [&](int j, int par) {
    uint2 h0, l0, h1, l1;
    wsplit2(act(xa[j][0].x, j), act(xa[j][0].y, j), h0.x, l0.x);
    wsplit2(act(xa[j][0].z, j), act(xa[j][0].w, j), h0.y, l0.y);
    wsplit2(act(xa[j][1].x, j), act(xa[j][1].y, j), h1.x, l1.x);
    wsplit2(act(xa[j][1].z, j), act(xa[j][1].w, j), h1.y, l1.y);
    _Float16* d = lds + par * BUF_HALFS + (row + 32 * j) * WP_STR + oct * 4;
    *reinterpret_cast<uint2*>(d) = h0;
    *reinterpret_cast<uint2*>(d + 32) = h1;
    *reinterpret_cast<uint2*>(d + TM * WP_STR) = l0;
    *reinterpret_cast<uint2*>(d + TM * WP_STR + 32) = l1;
  };
  auto commit_d = [&](int j, int par) {
    uint2 h0, l0, h1, l1;
    wsplit2(xd[j][0].x, xd[j][0].y, h0.x, l0.x);
    wsplit2(xd[j][0].z, xd[j][0].w, h0.y, l0.y);
    wsplit2(xd[j][1].x, xd[j][1].y, h1.x, l1.x);
    wsplit2(xd[j][1].z, xd[j][1].w, h1.y, l1.y);
    _Float16* d = lds + par * BUF_HALFS + A_HALFS + (row + 32 * j) * WP_STR + oct * 4;
    *reinterpret_cast<uint2*>(d) = h0;
    *reinterpret_cast<uint2*>(d + 32) = h1;
    *reinterpret_cast<uint2*>(d + TN * WP_STR) = l0;
    *reinterpret_cast<uint2*>(d + TN * WP_STR + 32) = l1;
  };

  wf32x16 acc_hi[MI][NJ], acc_lo[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc_hi[i][j][r] = 0.f;
        acc_lo[i][j][r] = 0.f;
      }

  if (s0 < s1) {  // stage s0 into LDS buffer 0, stage s0 + 1 into registers
    ln = s0 / spi;
    lq = s0 - ln * spi;
    lst = s0;
#pragma unroll
    for (int j = 0; j < NA; ++j) load_a(j);
#pragma unroll
    for (int j = 0; j < ND; ++j) load_d(j);
#pragma unroll
    for (int j = 0; j < NA; ++j) commit_a(j, 0);
#pragma unroll
    for (int j = 0; j < ND; ++j) commit_d(j, 0);
    advance();
#pragma unroll
    for (int j = 0; j < NA; ++j) load_a(j);
#pragma unroll
    for (int j = 0; j < ND; ++j) load_d(j);
    advance();
  }
  __syncthreads();

  const _Float16* a_lane = lds + (wci * 32 * MI + l31) * WP_STR + half * 8;
  const _Float16* d_lane = lds + A_HALFS + (wco * 32 * NJ + l31) * WP_STR + half * 8;
  for (int s = s0; s < s1; ++s) {
    const int par = (s - s0) & 1;
    const _Float16* al = a_lane + par * BUF_HALFS;
    const _Float16* dl = d_lane + par * BUF_HALFS;
    // operands one k-step ahead in a register double buffer (as in the 3x3 kernel: the reads of k-step kk + 1 precede
    // kk's staging writes in program order, so kk's MFMAs wait for registers only)
    whalf8 fa[2][MI][2], fb[2][NJ][2];
    auto frags = [&](int kk, int fp) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        fa[fp][i][0] = *reinterpret_cast<const whalf8*>(al + i * 32 * WP_STR + kk * 16);
        fa[fp][i][1] = *reinterpret_cast<const whalf8*>(al + (TM + i * 32) * WP_STR + kk * 16);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        fb[fp][j][0] = *reinterpret_cast<const whalf8*>(dl + j * 32 * WP_STR + kk * 16);
        fb[fp][j][1] = *reinterpret_cast<const whalf8*>(dl + (TN + j * 32) * WP_STR + kk * 16);
      }
    };
    frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      __builtin_amdgcn_sched_barrier(0);
      if (kk < 3) frags(kk + 1, (kk + 1) & 1);
      // the next stage's items (held in registers) -> the other buffer, NI / 4 per k-step; each item's registers are
      // refilled for the stage after it at once
#pragma unroll
      for (int it = kk * (NI / 4); it < (kk + 1) * (NI / 4); ++it) {
        if (it < NA) {
          commit_a(it, par ^ 1);
          load_a(it);
        } else {
          commit_d(it - NA, par ^ 1);
          load_d(it - NA);
        }
      }
      const int fp = kk & 1;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc_hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fp][i][0], fb[fp][j][0], acc_hi[i][j], 0, 0, 0);
          acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fp][i][0], fb[fp][j][1], acc_lo[i][j], 0, 0, 0);
          acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[fp][i][1], fb[fp][j][0], acc_lo[i][j], 0, 0, 0);
        }
#pragma unroll
      for (int m = 0; m < 3 * MI * NJ; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    advance();
    __syncthreads();
  }

  // epilogue: D[ci][co = l31] of every tile pair -> this run's slab [ci][co]
  float* wsb = p.ws + (size_t)blockIdx.x * p.cin_pad * p.cout_pad;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int co = co0 + (wco * NJ + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + (wci * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        wsb[(size_t)ci * p.cout_pad + co] = acc_hi[i][j][r] + acc_lo[i][j][r] * (1.0f / 2048.0f);
      }
    }
}

static bool wgrad_h2_pw_eligible(int cin, int c0, int c1, int cout, int ks, int stride, int ups, int plane) {
  (void)c0; (void)c1;
  return g_wgrad_h2 && ks == 1 && stride == 1 && ups == 0 && cin % 64 == 0 && cout % 64 == 0 && plane % 64 == 0;
}

// (tile, runs) of the pointwise split kernel: 128 x 128 tiles where both channel counts allow, about one workgroup per
// CU in all, at least four stages per run
static void wgrad_h2_pw_plan(int cin, int cout, int n, int plane, int* big, int* runs) {
  *big = (cin % 128 == 0 && cout % 128 == 0) ? 1 : 0;
  const int t = *big ? 128 : 64;
  const int pairs = (cin / t) * (cout / t);
  const int total = n * (plane / 64);
  int r = std::max(1, std::min(std::max(1, (*big ? 256 : 512) / pairs), std::max(1, total / 4)));  // (64 x 64 tiles: two per CU)
  if (r > 8) r = r / 8 * 8;  // (a multiple of the 8 XCDs, never more workgroups than CUs)
  *runs = std::min(r, std::max(1, total));
}

static int launch_wgrad_h2_pw(WgradP p, size_t ws_bytes, hipStream_t st) {
  int big, runs;
  const int plane = p.hin * p.win;
  wgrad_h2_pw_plan(p.cin, p.cout, p.n, plane, &big, &runs);
  const int t = big ? 128 : 64;
  p.ci_blocks = p.cin / t;
  const int pairs = p.ci_blocks * (p.cout / t);
  p.cin_pad = p.cin;
  p.cout_pad = p.cout;
  const size_t need = (size_t)runs * p.cin_pad * p.cout_pad * sizeof(float);
  if (p.ws == nullptr || ws_bytes < need)
    return fail(DSG_ERR_WORKSPACE_TOO_SMALL, "dsg_conv2d_wgrad: workspace %zu bytes < required %zu", ws_bytes, need);
  static bool raised = false;
  if (!raised) {
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_h2_pw_kernel<2, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_h2_pw_kernel<1, 1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised = true;
  }
  int pi = -1;
  if (prof_on())
    pi = prof_begin(5, 2.0 * p.n * plane * (double)p.cout * p.cin,
                    4.0 * ((double)p.n * p.cin * plane + (double)p.n * p.cout * plane), st);
  const size_t lds = (size_t)2 * 2 * 2 * t * WP_STR * 2;  // 2 buffers x (A + dY) x 2 pieces x t rows x 72 halfs x 2 bytes
  if (big) hipLaunchKernelGGL((conv_wgrad_h2_pw_kernel<2, 2>), dim3(runs, pairs), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((conv_wgrad_h2_pw_kernel<1, 1>), dim3(runs, pairs), dim3(256), lds, st, p);
  DSG_LAUNCH_CHECK();
  launch_wgrad_reduce(p.ws, runs, 1, p.cin, p.cout, p.cin_pad, p.cout_pad, p.dw, st);
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient of the mixed-precision tape: x and dY are channel-blocked 16-bit tensors [N][C/8][H][W][8] (bf16 or
// fp16), one MFMA per product, fp32 accumulate.
//
// K is the pixel index, but in memory a pixel's 8 channels are adjacent -- the operands arrive transposed.  The LDS
// keeps the global layout ([pixel][32 channels], 64 bytes per pixel: written with the 16-byte pieces as they were
// loaded) and the fragments are fetched with ds_read_b64_tr_b16, gfx950's transposing read: within a 16-lane group
// source lane s supplies the address of 4 contiguous 16-bit values and lane L receives, in slot k, element (L & 3) of
// source lane 4 k + (L >> 2) (measured: tools/probes/probe_tr.hip).  With source lane s pointing at
// [pixel p0 + (s >> 2)][channel quad (s & 3)] of a 16-channel group, lane L ends up with channel L of pixels
// p0 .. p0 + 3: two reads give the 8 k-values of one 32x32x16 operand row, for A (activations, rows = ci) and B (dY,
// columns = co) alike, so the pixel order inside a k-step is the same on both sides.  A tap's (dy, dx) shift is just
// another pixel address: no pre-shifted copies.  64-byte pixel rows put the 4 pixels of a read on 4 x 16 distinct banks.
//
// Workgroup = 64 ci x 64 co x 9 taps (wave w: ci tile w >> 1, co tile w & 1, 9 accumulator tiles), walking consecutive
// row pairs of one 32-column strip of one image through a 6-row LDS ring (every input row is activated and converted
// once); GroupNorm affine + SiLU are recomputed in fp32 from the saved pre-norm tensor; split-K partials go to the same
// workspace / fixed-order reduce as the fp32 kernels.
// ---------------------------------------------------------------------------------------------------
struct Wgrad16P {
  const void* src0;
  const void* src1;
  int c0, c1, cin;
  int n, h, w;          // stride 1, padding 1: source and dY maps have the same size
  int cout;
  const void* dy;       // blocked [N][dy_ctotal/8][h][w][8]; this conv's channels start at dy_coff
  int dy_ctotal, dy_coff;
  const float* ss;      // optional [N][cin][2]
  int silu;
  float* ws;            // [slab][9][cin][cout]
  int tiles_x, stages;  // 32-column strips per row, row pairs per image
  int ci_blocks, nrs;   // 64-channel ci blocks; row splits per strip
  int spw;              // strips per workgroup (> 1 only with nrs == 1): consecutive strips accumulate into ONE partial slab
                        // -- at 512 channels the (ci, co) block pairs alone fill the chip, and a slab per image meant 300 MB
                        // of partials for the reduce pass to read back
  float* dysum_ws;      // optional [slab][cout]: per-run sums of dY over its pixels (bias / time-embedding gradients):
                        // the dY tiles pass through this kernel anyway -- saves a pass of its own over dY
  // Sampler convs (round 6): the kernel's K grid (h x w) stays the FULL-resolution map, one operand lives at half resolution
  // and is addressed through a shift -- no materialised copy:
  //   xsh = 1  Upsample2D + conv (nearest x2 in front of the 3x3): x is [N][C/8][h/2][w/2][8], pixel (y, x) reads (y >> 1, x >> 1)
  //            (was: dsg_upsample_nearest2x_blocked into a 4x larger tensor, 0.26 ms per layer at batch 128, then read back)
  //   dsh = 1  Downsample2D's stride-2 conv: dY is [N][Co/8][h/2][w/2][8]; pixel (y, x) of the K grid is dY[y/2][x/2] when both are
  //            even and ZERO otherwise (was: a strided torch copy into a zeroed full-resolution buffer per step)
  int xsh, dsh;
  // Upsample2D + conv, folded (FOLD = 1): the K grid is the LOW-resolution map (h x w = x's), dY [N][Co/8][2h][2w][8] is read as its
  // space-to-depth image with 4 Co channels in parity-major order (channel p * fold_co + c = dY[c] at pixels (2y + py, 2x + px),
  // p = 2 py + px; cout = 4 fold_co), and a workgroup -- whose co block lies inside ONE parity -- contracts only the 2 x 2 taps of the
  // 3 x 3 window that parity reads: x rows y + py - 1 + {0, 1}.  16 (tap, parity) products per low-resolution pixel instead of the 36
  // of nine taps at full resolution; wgrad_fold_reduce_kernel adds the four (tap, parity) terms each 3 x 3 weight gradient is the sum of.
  int fold_co;
};

typedef short wg_s4 __attribute__((ext_vector_type(4)));
constexpr int W16_SLOTS = 6;
// The same matrix instruction with its accumulator pinned to ARCHITECTURAL registers.  A kernel compiled for 512 registers
// gets the AGPR form of every MFMA builtin: accumulators must sit in the 256 AGPRs, and the 18 tiles (288 registers) of the
// wide weight-gradient workgroup made hipcc shuttle tiles between the files (880 v_accvgpr moves per 72 MFMAs).  Sixteen
// tiles stay with the builtin, two go through here.  The hazard recogniser does not see inside: callers keep any other
// access to `c` at least one k-step (16 passes) away.
// a copy the register allocator cannot fold away: the source registers are free for the next loads from here on
__device__ __forceinline__ uint4 pinned_copy(const uint4& v) {
  uint4 r;
  asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
               : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w)
               : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
  return r;
}
template <int PREC>
__device__ __forceinline__ void mma16_vgpr_acc(const half8& a, const half8& b, wf32x16& c) {
  if constexpr (PREC == 1) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int KS, int COT = 1>
struct W16Geom {
  static constexpr int PW = 32 + 2 * (KS / 2);                  // patch columns (halo of KS / 2)
  static constexpr int A_HALFS = 2 * W16_SLOTS * PW * 32;       // [ci tile 2][slot][col][32 ch]
  static constexpr int D_HALFS = 2 * COT * 64 * 32;             // [co tile 2 COT][px 64][32 co], double-buffered
  static constexpr int LDS_BYTES = (A_HALFS + 2 * D_HALFS) * 2 + (COT == 2 ? 256 * 16 : 0);  // (+ a 16-byte sink per thread)
};

// KS = 3: 3x3, padding 1.  KS = 1: pointwise (shortcuts, attention projections): the map is re-tiled by the host as
// h * w / 32 rows of 32 pixels (a channel-blocked tensor is linear in the pixel index), no halo, one tap.
// ACT: 1 = the sources go through GroupNorm affine + SiLU (every resnet conv), 0 = used as they are (shortcuts, samplers),
// 2 = decided at run time (affine without SiLU: the attention projections) -- compile-time in the two common cases: the
// run-time flags cost branches in the staging code and registers the kernel does not have
// COT = 2 (cout % 128 == 0): the workgroup covers 64 ci x 128 co and a wave keeps TWO co tiles of its ci tile -- 18
// accumulator tiles, 512 registers, one workgroup per CU.  An activation fragment then feeds two matrix instructions (11
// LDS fragments per 18 instead of 10 per 9: at COT = 1 the LDS pipe is as busy as the matrix pipe), the activation
// arithmetic of the staging is paid once per 128 output channels instead of per 64, and x is re-read cout / 128 times.
template <int PREC, int KS, int ACT = 2, int COT = 1, int FOLD = 0>
__global__ __launch_bounds__(256, COT == 2 ? 1 : 2) void conv_wgrad16_kernel(Wgrad16P p) {
  static_assert(!FOLD || (KS == 3 && ACT == 0), "folded up-sampler form: 3x3, sources used as they are");
  constexpr int W16_PW = W16Geom<KS>::PW, W16_A_HALFS = W16Geom<KS>::A_HALFS, PADK = KS / 2, TAPS = FOLD ? 4 : KS * KS;
  constexpr int W16_D_HALFS = W16Geom<KS, COT>::D_HALFS, COW = 64 * COT, DCB = 8 * COT, DSH = COT == 2 ? 4 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm16[];
  unsigned short* Ab = reinterpret_cast<unsigned short*>(wsm16);
  unsigned short* Db = Ab + W16_A_HALFS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int cit = wave >> 1, cot = (wave & 1) * COT;  // (the wave's first co tile)

  int pair_id, slab_id;
  wgrad_xcd_ids(pair_id, slab_id, p.spw == 1 ? p.nrs : 0, p.tiles_x);
  const int cib = pair_id % p.ci_blocks, cob = pair_id / p.ci_blocks;
  const int ci0 = cib * 64, co0 = cob * COW;
  const int plane = p.h * p.w;
  // half-resolution operands of the sampler convs (uniform shifts: see Wgrad16P)
  const int xsh = __builtin_amdgcn_readfirstlane(p.xsh), dsh = __builtin_amdgcn_readfirstlane(p.dsh);
  const int xplane = plane >> (2 * xsh), xw = p.w >> xsh;
  const int dplane = FOLD ? 4 * plane : plane >> (2 * dsh), dw_ = FOLD ? 2 * p.w : p.w >> dsh;
  // folded form: this co block's pixel parity (uniform) and its first channel inside dY
  const int fpp = FOLD ? co0 / p.fold_co : 0, fpy = fpp >> 1, fpx = fpp & 1;
  const int co0d = FOLD ? co0 - fpp * p.fold_co : co0;
  const bool has_ss = ACT == 2 ? p.ss != nullptr : ACT == 1;
  const bool do_silu = ACT == 2 ? (has_ss && p.silu) : ACT == 1;

  // this workgroup's runs: strips [run * spw, (run + 1) * spw) (image n, column tile tx), stages [s0, s1) of each
  const int run = slab_id / p.nrs, rs = slab_id - run * p.nrs;
  const int per = (p.stages + p.nrs - 1) / p.nrs;
  const int s0 = rs * per, s1 = min(p.stages, s0 + per);
  int n = 0, ox0 = 0;
  const unsigned short* xsrc = nullptr;
  const unsigned short* dsrc = nullptr;
  const bool in0 = ci0 < p.c0;  // the 64-channel ci block sits entirely in one of the two concatenated sources (c0 % 64 == 0)

  // staging items.  A: (row of the pair, column 0..33, channel block 0..7) = 544, three rounds; dY: (pixel 0..63, co block) = 512 COT
  const int a_cb = tid & 7;  // the same channel block in every round: its scale / shift live in registers
  float sc[8], sh[8];
  auto begin_strip = [&](int strip) {  // per-strip state: image, column tile, source bases, the image's scale / shift
    n = strip / p.tiles_x;
    ox0 = (strip - n * p.tiles_x) * 32;
    xsrc = in0 ? static_cast<const unsigned short*>(p.src0) + ((size_t)n * p.c0 + ci0) * xplane
               : static_cast<const unsigned short*>(p.src1) + ((size_t)n * p.c1 + (ci0 - p.c0)) * xplane;
    dsrc = static_cast<const unsigned short*>(p.dy) + ((size_t)n * p.dy_ctotal + p.dy_coff + co0d) * dplane;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = 1.f;
      sh[j] = 0.f;
      if (has_ss) {
        const float2 t2 = *reinterpret_cast<const float2*>(p.ss + ((size_t)n * p.cin + ci0 + a_cb * 8 + j) * 2);
        sc[j] = t2.x;
        sh[j] = t2.y;
      }
    }
  };
  int a_row[3], a_col[3];
  bool a_use[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int pc = (tid + 256 * u) >> 3;  // 0..95; valid below 68
    a_use[u] = pc < 2 * W16_PW;
    a_row[u] = pc / W16_PW;
    a_col[u] = pc - a_row[u] * W16_PW;
  }
  uint4 xa[3], xd[2 * COT];
  bool va[3];
  auto load_rows_to = [&](int k, uint4 (&dst)[3], bool (&valid)[3]) {  // input rows 2k - PADK, 2k + 1 - PADK of the strip (columns ox0 - PADK ...)
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int y = 2 * k - PADK + a_row[u], x = ox0 - PADK + a_col[u];
      valid[u] = a_use[u] && (unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w;
      const size_t off = (size_t)a_cb * 8 * xplane + ((size_t)((valid[u] ? y : 0) >> xsh) * xw + ((valid[u] ? x : 0) >> xsh)) * 8;
      dst[u] = *reinterpret_cast<const uint4*>(xsrc + off);
    }
  };
  auto load_rows = [&](int k) { load_rows_to(k, xa, va); };
  auto load_dy_to = [&](int s, uint4 (&dst)[2 * COT]) {  // output rows 2s, 2s+1 (clamped past the image: never used)
#pragma unroll
    for (int u = 0; u < 2 * COT; ++u) {
      const int id = tid + 256 * u, cb = id & (DCB - 1), px = id >> DSH;
      const int y = min(2 * s + (px >> 5), p.h - 1), x = ox0 + (px & 31);
      const size_t dpx = FOLD ? (size_t)(2 * y + fpy) * dw_ + (2 * x + fpx) : (size_t)(y >> dsh) * dw_ + (x >> dsh);
      const uint4 q = *reinterpret_cast<const uint4*>(dsrc + (size_t)cb * 8 * dplane + dpx * 8);
      const bool hole = !FOLD && dsh != 0 && ((y | x) & 1) != 0;   // (stride-2 conv: the odd rows / columns of the K grid carry no dY)
      dst[u] = hole ? make_uint4(0u, 0u, 0u, 0u) : q;
    }
  };
  auto load_dy = [&](int s) { load_dy_to(s, xd); };
  // staging of a row pair in pieces (the wide workgroup deals them over the k loop: one wave per SIMD, nobody else's
  // matrix instructions to hide the arithmetic behind): act_piece = two channels of round u, write_rows = round u's 16 bytes
  unsigned oa[3][4];
  uint4 ca[3], cd[2 * COT];  // the pair / dY tile being staged (COT 2: pinned copies, xa / xd already take the next loads)
  bool cva[3];
  auto act_piece = [&](int u, int jp) {
    const unsigned w = jp == 0 ? ca[u].x : jp == 1 ? ca[u].y : jp == 2 ? ca[u].z : ca[u].w;
#if defined(DSG_W16_ABL_NOSTAGE)
    oa[u][jp] = cva[u] ? w : 0u;
    return;
#endif
    float a = lo16<PREC>(w), b = hi16<PREC>(w);
    if (has_ss) {
      a = a * sc[2 * jp] + sh[2 * jp];
      b = b * sc[2 * jp + 1] + sh[2 * jp + 1];
    }
    if (do_silu) {
      a = silu_fast_b(a);
      b = silu_fast_b(b);
    }
    oa[u][jp] = cva[u] ? pack2<PREC>(a, b) : 0u;  // zero padding applies to the ACTIVATED map
  };
  // the same arithmetic in three phases, each dealt to a later slot (wide workgroup): with one wave per SIMD the chain
  // unpack -> affine -> exp -> 1 + e -> rcp -> x * r -> pack of ONE piece issues as a string of dependent instructions
  // (and hazard nops behind the two transcendentals); three pieces in flight give every slot independent work
  float pa[12], pb[12], ea[12], eb[12];
  auto piece_p0 = [&](int q) {
    const int u = q >> 2, jp = q & 3;
    const unsigned w = jp == 0 ? ca[u].x : jp == 1 ? ca[u].y : jp == 2 ? ca[u].z : ca[u].w;
#if defined(DSG_W16_ABL_NOSTAGE)
    oa[u][jp] = cva[u] ? w : 0u;
    return;
#endif
    float a = lo16<PREC>(w), b = hi16<PREC>(w);
    if (has_ss) {
      a = a * sc[2 * jp] + sh[2 * jp];
      b = b * sc[2 * jp + 1] + sh[2 * jp + 1];
    }
    pa[q] = a;
    pb[q] = b;
    if (do_silu) {
      ea[q] = __expf(-a);
      eb[q] = __expf(-b);
    }
  };
  auto piece_p1 = [&](int q) {
#if defined(DSG_W16_ABL_NOSTAGE)
    return;
#endif
    if (do_silu) {
      ea[q] = __builtin_amdgcn_rcpf(1.0f + ea[q]);
      eb[q] = __builtin_amdgcn_rcpf(1.0f + eb[q]);
    }
  };
  auto piece_p2 = [&](int q) {
#if defined(DSG_W16_ABL_NOSTAGE)
    return;
#endif
    const int u = q >> 2, jp = q & 3;
    float a = pa[q], b = pb[q];
    if (do_silu) {
      a *= ea[q];
      b *= eb[q];
    }
    oa[u][jp] = pack2<PREC>(a, b) & (cva[u] ? ~0u : 0u);  // (an AND, not a select: hipcc turns a select between 0 and the
                                                          // end of a chain with transcendentals into a branch and sinks the
                                                          // whole chain -- all three phases -- into it)
  };
  auto write_rows = [&](int k, int u) {  // -> ring slots (2k) % 6, (2k) % 6 + 1
    const int slot = (2 * k) % W16_SLOTS + a_row[u];
    unsigned short* dst = Ab + (a_cb >> 2) * (W16_SLOTS * W16_PW * 32) + (slot * W16_PW + a_col[u]) * 32 + (a_cb & 3) * 8;
    if constexpr (COT == 2) {  // no branch (the pieces would sink into it, back into one lump): idle threads write to their sink
      if (!a_use[u]) dst = Db + 2 * W16_D_HALFS + tid * 8;
    } else {
      if (!a_use[u]) return;
    }
    *reinterpret_cast<uint4*>(dst) = make_uint4(oa[u][0], oa[u][1], oa[u][2], oa[u][3]);
  };
  auto commit_rows = [&](int k) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      ca[u] = xa[u];
      cva[u] = va[u];
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) act_piece(u, jp);
      write_rows(k, u);
    }
  };
  const bool want_dysum = p.dysum_ws != nullptr && cib == 0;  // (one ci block per co block owns the by-product)
  float dsum[8];  // this thread's channel block (tid & (DCB - 1)) of dY, summed over its pixels of the run
#pragma unroll
  for (int j = 0; j < 8; ++j) dsum[j] = 0.f;
  auto commit_dy_u = [&](int par, bool count, int u) {
    const int id = tid + 256 * u, cb = id & (DCB - 1), px = id >> DSH;
    unsigned short* dst = Db + par * W16_D_HALFS + (cb >> 2) * (64 * 32) + px * 32 + (cb & 3) * 8;
    *reinterpret_cast<uint4*>(dst) = cd[u];
    if (want_dysum && count) {
      const unsigned w4[4] = {cd[u].x, cd[u].y, cd[u].z, cd[u].w};
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        dsum[2 * jp] += lo16<PREC>(w4[jp]);
        dsum[2 * jp + 1] += hi16<PREC>(w4[jp]);
      }
    }
  };
  auto commit_dy = [&](int par, bool count) {
#pragma unroll
    for (int u = 0; u < 2 * COT; ++u) {
      cd[u] = xd[u];
      commit_dy_u(par, count, u);
    }
  };

  wf32x16 acc[COT][TAPS];
#pragma unroll
  for (int ct = 0; ct < COT; ++ct)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][t][r] = 0.f;
  wf32x16 accv[COT];  // COT 2: tap 0 of both co tiles, in architectural registers (mma16_vgpr_acc)
#pragma unroll
  for (int ct = 0; ct < COT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) accv[ct][r] = 0.f;

  // transposing-read addressing of this lane: source lane s = lane & 15 -> pixel + (s >> 2), channel quad (s & 3) of the
  // 16-channel group (lane >> 4) & 1 of the wave's 32-channel tile
  const int s16 = lane & 15, g16 = (lane >> 4) & 1;
  const int t_px = s16 >> 2, t_ch = g16 * 16 + (s16 & 3) * 4;
  // (the lane's pixel offset inside a k-step folded into the bases: what is left per fragment is a compile-time offset)
  // (folded form: the parity's 2 x 2 corner of the window starts at column fpx / row fpy of it)
  const unsigned short* a_lane = Ab + cit * (W16_SLOTS * W16_PW * 32) + t_ch + (half * 8 + t_px + fpx) * 32;
  const unsigned short* d_lane = Db + cot * (64 * 32) + t_ch + (half * 8 + t_px) * 32;
  auto tr4 = [](const unsigned short* q) -> wg_s4 {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) wg_s4*)(const_cast<unsigned short*>(q)));
  };
  typedef short wg_s8 __attribute__((ext_vector_type(8)));

  for (int it = 0; it < p.spw; ++it) {  // (body not re-indented)
  const int strip = run * p.spw + it;
  begin_strip(strip);
  // prologue: row pairs s0, s0+1 and dY(s0) in LDS; pair s0+2 and dY(s0+1) in registers
  if (s0 < s1) {
    load_rows(s0);
    load_dy(s0);
    commit_rows(s0);
    commit_dy(0, true);
    load_rows(s0 + 1);
    commit_rows(s0 + 1);
    load_rows(s0 + 2);
    load_dy(s0 + 1);
  }
  __syncthreads();

  for (int s = s0; s < s1; ++s) {
    const int par = (s - s0) & 1;
    // stage the next pair / dY tile (other ring slots, other dY buffer), then fetch the ones after them
    if constexpr (COT == 1) {
      commit_rows(s + 2);
      commit_dy(par ^ 1, s + 1 < s1);  // (the tile after the run's last one is loaded clamped and never used)
      load_rows(s + 3);
      load_dy(s + 2);
    }
    const unsigned short* dl = d_lane + par * W16_D_HALFS;
    const unsigned short* arow[KS + 1];  // ring rows of input rows 2s - PADK + j: one address per row and stage
#pragma unroll
    for (int j = 0; j <= KS; ++j) arow[j] = a_lane + ((2 * s + j + fpy) % W16_SLOTS) * (W16_PW * 32);
    // Fragments are fetched ONE K-STEP AHEAD, each into the registers its MFMA has just read: a tap's operand is in flight
    // for the nine MFMAs of a k-step instead of being waited for right behind its read (left to itself the compiler issued
    // most reads directly in front of their MFMA: an LDS round trip per matrix instruction).
    auto b_frag = [&](int kk, int ct) -> half8 {
#if defined(DSG_W16_ABL_NOREAD)
      return __builtin_bit_cast(half8, wg_s8{(short)kk, (short)ct, 2, 3, 4, 5, 6, 7});
#endif
      const int orow = kk >> 1, colb = (kk & 1) * 16;  // (+ this lane's first pixel of the k-step: in d_lane)
      const wg_s4 b0 = tr4(dl + ct * (64 * 32) + (orow * 32 + colb) * 32), b1 = tr4(dl + ct * (64 * 32) + (orow * 32 + colb + 4) * 32);
      return __builtin_bit_cast(half8, wg_s8{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w});
    };
    auto a_frag = [&](int kk, int tp) -> half8 {
#if defined(DSG_W16_ABL_NOREAD)
      return __builtin_bit_cast(half8, wg_s8{(short)kk, (short)tp, 2, 3, 4, 5, 6, 7});
#endif
      const int orow = kk >> 1, colb = (kk & 1) * 16;
      const int dy = FOLD ? tp >> 1 : tp / KS, dx = FOLD ? tp & 1 : tp % KS;
      const unsigned short* ap = arow[orow + dy] + (colb + dx) * 32;  // input row 2s - PADK + orow + dy (+ fpy)
      const wg_s4 a0 = tr4(ap), a1 = tr4(ap + 4 * 32);
      return __builtin_bit_cast(half8, wg_s8{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w});
    };
    // the stage's 4 * TAPS (k-step, tap) products in one sequence; operand g + AHEAD is fetched right behind MFMA g, into
    // the ring slot MFMA g - 1 has just read (the LDS counter is four bits wide: a whole k-step of reads in flight --
    // 18 -- made every wait a wait for all of them)
#if defined(DSG_W16_AHEAD)  // (tools/ timing experiments)
    constexpr int NG = 4 * TAPS, AHEAD = COT == 2 ? DSG_W16_AHEAD : (TAPS >= 3 ? 3 : 1), RING = AHEAD + 1;
#else
    constexpr int NG = 4 * TAPS, AHEAD = TAPS >= 3 ? 3 : 1, RING = AHEAD + 1;
#endif
    half8 fa[RING], fb[2][COT];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) fb[0][ct] = b_frag(0, ct);
#pragma unroll
    for (int g = 0; g < AHEAD && g < NG; ++g) fa[g % RING] = a_frag(g / TAPS, g % TAPS);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int kk = g / TAPS, tp = g % TAPS;
      if constexpr (COT == 2) {  // this slot's share of the staging (two matrix instructions = 64 cycles per slot)
        if (g == 0) {  // the loads of a whole stage ahead go out first, into the registers the pinned copies have freed
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            ca[u] = pinned_copy(xa[u]);
            cva[u] = va[u];
          }
#pragma unroll
          for (int u = 0; u < 2 * COT; ++u) cd[u] = pinned_copy(xd[u]);
          load_rows(s + 3);
          load_dy(s + 2);
        } else if (g <= 14) {
          if (g - 1 < 12) piece_p0(g - 1);
          if (g >= 2 && g - 2 < 12) piece_p1(g - 2);
          if (g >= 3) {
            piece_p2(g - 3);
            if (((g - 3) & 3) == 3) write_rows(s + 2, (g - 3) >> 2);
          }
        } else if (!FOLD && g <= 18) {
          commit_dy_u(par ^ 1, s + 1 < s1, g - 15);
        }
        if constexpr (FOLD) {  // (16 slots per stage instead of 36: the dY tile goes out next to the last pieces)
          if (g >= 12) commit_dy_u(par ^ 1, s + 1 < s1, g - 12);
        }
      }
#if defined(DSG_W16_ABL_NOMMA)   // (tools/ timing experiments only: wrong results, the loop's time without a component)
      asm volatile("" ::"v"(fa[g % RING]), "v"(fb[kk & 1][0]));
#else
#pragma unroll
      for (int ct = 0; ct < COT; ++ct) {
        if (COT == 2 && tp == 0) mma16_vgpr_acc<PREC>(fa[g % RING], fb[kk & 1][ct], accv[ct]);  // (16 more MFMAs before anything else touches it)
        else acc[ct][tp] = mma16<PREC>(fa[g % RING], fb[kk & 1][ct], acc[ct][tp]);
      }
#endif
      if (g + AHEAD < NG) fa[(g + AHEAD) % RING] = a_frag((g + AHEAD) / TAPS, (g + AHEAD) % TAPS);
      if (kk < 3) {  // the next k-step's dY fragments: one per product near the end of this k-step
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
          if (tp == (TAPS > 1 ? TAPS - 1 - COT + ct : 0)) fb[(kk + 1) & 1][ct] = b_frag(kk + 1, ct);
      }
      // pin the order of matrix instructions and LDS reads (everything else -- the staging arithmetic, its loads and LDS
      // writes -- may still be moved between them): left alone the scheduler sinks each read to its use
      if constexpr (COT == 2) __builtin_amdgcn_sched_barrier(0);  // (the dealt staging stays in its slot)
      else __builtin_amdgcn_sched_barrier(0x216);
    }
    __syncthreads();
  }

  if (want_dysum) {  // 32 / COT threads share a channel block: fixed-order sum through LDS (the strip's K loop is done with it)
    float* red = reinterpret_cast<float*>(wsm16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[tid * 8 + j] = dsum[j];
      dsum[j] = 0.f;
    }
    __syncthreads();
    if (tid < COW) {
      const int cb = tid >> 3, j = tid & 7;
      float t = 0.f;
      for (int k = 0; k < 256 / DCB; ++k) t += red[(cb + DCB * k) * 8 + j];
      p.dysum_ws[((size_t)strip * p.nrs + rs) * p.cout + co0 + tid] = t;  // (per strip: the sums are per image)
    }
    __syncthreads();  // the next strip's prologue writes the same LDS
  }
  }  // strips of this workgroup
  // epilogue: D[ci rows][co = l31]; partials to this run's slab [tap][ci][co]
  float* wsb = p.ws + (size_t)slab_id * TAPS * p.cin * p.cout;
#pragma unroll
  for (int ct = 0; ct < COT; ++ct) {
    const int co = co0 + (cot + ct) * 32 + l31;
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + cit * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        wsb[((size_t)tp * p.cin + ci) * p.cout + co] = (COT == 2 && tp == 0) ? accv[ct][r] : acc[ct][tp][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// Pointwise (1x1) 16-bit weight gradient on tiles of its own (shortcuts, attention projections; tuning key 30).  With one tap a
// 64 x 64 workgroup of the kernel above has four MFMAs per wave between two barriers, two transposing reads per MFMA, and
// re-reads x cout / 64 and dY cin / 64 times: 115-295 us per layer for 26-52 GF, neither matrix- nor memory-bound.  Here a
// workgroup covers 64 MI ci x 64 NJ co (MI, NJ in {1, 2}: 128 x 128 where the channel counts allow), wave (wm, wn) owns the MI x NJ
// tiles (2 wm + i, 2 wn + j)/2 ... of it: per 16-pixel k-step MI + NJ fragments feed MI x NJ MFMAs, and x / dY are re-read
// cout / (64 NJ) / cin / (64 MI) times.  A stage is 64 consecutive pixels of one image ([N][C/8][H W][8] is linear in the pixel);
// LDS [buffer 2][tile][64 px][32 ch] for both operands, one barrier per stage, loads a stage ahead, fragments a k-step ahead.
// A run = consecutive stages of ONE image (the scale / shift table and the dY-sum by-product are per image).
// ---------------------------------------------------------------------------------------------------
struct Wgrad16PwP {
  const void* src0;
  const void* src1;
  int c0, c1, cin, n, plane, cout;
  const void* dy;
  int dy_ctotal, dy_coff;
  const float* ss;
  int silu;
  float* ws;        // [slab][cin][cout]
  int ci_blocks;    // cin / (64 MI)
  int rpi;          // runs per image
  float* dysum_ws;  // optional [slab][cout]
};

template <int PREC, int ACT, int MI, int NJ>
__global__ __launch_bounds__(256, 2) void conv_wgrad16_pw_kernel(Wgrad16PwP p) {
  constexpr int TM = 64 * MI, TN = 64 * NJ, AT = 2 * MI, DT = 2 * NJ;      // channels / 32-channel tiles per workgroup
  constexpr int TILE = 64 * 32;                                             // halfs of one [64 px][32 ch] tile
  constexpr int BUF = (AT + DT) * TILE;
  constexpr int AU = TM / 32, DU = TN / 32;                                 // 16-byte pieces per thread and stage
  constexpr int ACB = TM / 8, DCB = TN / 8;                                 // channel blocks per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char wsmpw[];
  unsigned short* L = reinterpret_cast<unsigned short*>(wsmpw);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31, wm = wave >> 1, wn = wave & 1;
  int pair_id, slab_id;
  wgrad_xcd_ids(pair_id, slab_id);
  const int cib = pair_id % p.ci_blocks, cob = pair_id / p.ci_blocks;
  const int ci0 = cib * TM, co0 = cob * TN;
  const int n = slab_id / p.rpi, run = slab_id - n * p.rpi;
  const int stages = p.plane / 64, per = (stages + p.rpi - 1) / p.rpi;
  const int s0 = run * per, s1 = min(stages, s0 + per);
  const bool has_ss = ACT == 2 ? p.ss != nullptr : ACT == 1;
  const bool do_silu = ACT == 2 ? (has_ss && p.silu) : ACT == 1;
  const bool in0 = ci0 < p.c0;  // (the ci block sits in one source: c0 % TM == 0)
  const unsigned short* xsrc = in0 ? static_cast<const unsigned short*>(p.src0) + ((size_t)n * p.c0 + ci0) * p.plane
                                   : static_cast<const unsigned short*>(p.src1) + ((size_t)n * p.c1 + (ci0 - p.c0)) * p.plane;
  const unsigned short* dsrc = static_cast<const unsigned short*>(p.dy) + ((size_t)n * p.dy_ctotal + p.dy_coff + co0) * p.plane;

  // staging items: piece id = tid + 256 u -> (pixel id / CB, channel block id % CB): the channel block is the same in every round
  const int a_cb = tid & (ACB - 1), d_cb = tid & (DCB - 1);
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = 1.f;
    sh[j] = 0.f;
    if (has_ss) {
      const float2 t2 = *reinterpret_cast<const float2*>(p.ss + ((size_t)n * p.cin + ci0 + a_cb * 8 + j) * 2);
      sc[j] = t2.x;
      sh[j] = t2.y;
    }
  }
  uint4 xa[AU], xd[DU];
  auto load_stage = [&](int s) {  // (clamped past the image: never used)
    const int px0 = min(s, stages - 1) * 64;
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const int px = (tid + 256 * u) / ACB;
      xa[u] = *reinterpret_cast<const uint4*>(xsrc + ((size_t)a_cb * p.plane + px0 + px) * 8);
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int px = (tid + 256 * u) / DCB;
      xd[u] = *reinterpret_cast<const uint4*>(dsrc + ((size_t)d_cb * p.plane + px0 + px) * 8);
    }
  };
  const bool want_dysum = p.dysum_ws != nullptr && cib == 0;
  float dsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dsum[j] = 0.f;
  auto commit_stage = [&](int par, bool count) {
    unsigned short* Ab = L + par * BUF;
    unsigned short* Db = Ab + AT * TILE;
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const int px = (tid + 256 * u) / ACB;
      const unsigned w4[4] = {xa[u].x, xa[u].y, xa[u].z, xa[u].w};
      unsigned o4[4];
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        float a = lo16<PREC>(w4[jp]), b = hi16<PREC>(w4[jp]);
        if (has_ss) {
          a = a * sc[2 * jp] + sh[2 * jp];
          b = b * sc[2 * jp + 1] + sh[2 * jp + 1];
        }
        if (do_silu) {
          a = silu_fast_b(a);
          b = silu_fast_b(b);
        }
        o4[jp] = (has_ss || do_silu) ? pack2<PREC>(a, b) : w4[jp];
      }
      *reinterpret_cast<uint4*>(Ab + (a_cb >> 2) * TILE + px * 32 + (a_cb & 3) * 8) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int px = (tid + 256 * u) / DCB;
      *reinterpret_cast<uint4*>(Db + (d_cb >> 2) * TILE + px * 32 + (d_cb & 3) * 8) = xd[u];
      if (want_dysum && count) {
        const unsigned w4[4] = {xd[u].x, xd[u].y, xd[u].z, xd[u].w};
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          dsum[2 * jp] += lo16<PREC>(w4[jp]);
          dsum[2 * jp + 1] += hi16<PREC>(w4[jp]);
        }
      }
    }
  };

  wf32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposing-read addressing (see conv_wgrad16_kernel): source lane s = lane & 15 -> pixel + (s >> 2), channel quad s & 3
  const int s16 = lane & 15, g16 = (lane >> 4) & 1;
  const int lane_off = g16 * 16 + (s16 & 3) * 4 + (half * 8 + (s16 >> 2)) * 32;
  auto tr4 = [](const unsigned short* q) -> wg_s4 {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(const_cast<unsigned short*>(q)));
  };
  typedef short wg_s8 __attribute__((ext_vector_type(8)));
  auto frag = [&](const unsigned short* tile, int kk) -> half8 {
    const wg_s4 v0 = tr4(tile + kk * 16 * 32), v1 = tr4(tile + (kk * 16 + 4) * 32);
    return __builtin_bit_cast(half8, wg_s8{v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w});
  };

  if (s0 < s1) {
    load_stage(s0);
    commit_stage(0, true);
    load_stage(s0 + 1);
  }
  __syncthreads();
  for (int s = s0; s < s1; ++s) {
    const int par = (s - s0) & 1;
    commit_stage(par ^ 1, s + 1 < s1);
    load_stage(s + 2);
    const unsigned short* at = L + par * BUF + (wm * MI) * TILE + lane_off;
    const unsigned short* dt = L + par * BUF + (AT + wn * NJ) * TILE + lane_off;
    half8 fa[2][MI], fb[2][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[0][i] = frag(at + i * TILE, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[0][j] = frag(dt + j * TILE, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[(kk + 1) & 1][i] = frag(at + i * TILE, kk + 1);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[(kk + 1) & 1][j] = frag(dt + j * TILE, kk + 1);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mma16<PREC>(fa[kk & 1][i], fb[kk & 1][j], acc[i][j]);
    }
    __syncthreads();
  }

  if (want_dysum) {  // 256 / DCB threads share a channel block: fixed-order sum through LDS
    float* red = reinterpret_cast<float*>(wsmpw);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = dsum[j];
    __syncthreads();
    if (tid < TN) {
      const int cb = tid >> 3, j = tid & 7;
      float t = 0.f;
      for (int k = 0; k < 256 / DCB; ++k) t += red[(cb + DCB * k) * 8 + j];
      p.dysum_ws[(size_t)slab_id * p.cout + co0 + tid] = t;
    }
  }
  float* wsb = p.ws + (size_t)slab_id * p.cin * p.cout;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int co = co0 + (wn * NJ + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        wsb[(size_t)ci * p.cout + co] = acc[i][j][r];
      }
    }
}

static int g_wgrad16_pw = 1;  // dsg_set_tuning key 30 (tests / A-B runs): 0 = the 3x3 kernel's one-tap instantiation
void conv_wgrad16_set_pw(int v) { g_wgrad16_pw = v; }
// tile multiples of the pointwise kernel for these channel counts, and its runs per image (>= 512 workgroups where the map allows)
static void wgrad16_pw_plan(int c0, int c1, int cout, int n, int plane, int* mi, int* nj, int* rpi) {
  const int cin = c0 + c1;
  *mi = (cin % 128 == 0 && (c1 == 0 || c0 % 128 == 0)) ? 2 : 1;
  *nj = cout % 128 == 0 ? 2 : 1;
  const int pairs = (cin / (64 * *mi)) * (cout / (64 * *nj)), stages = plane / 64;
  int r = std::max(1, std::min(stages, cdiv(512, pairs * n)));
  while (stages % r != 0) ++r;  // (equal runs)
  *rpi = r;
}

static int g_wgrad16_wide = 1;  // dsg_set_tuning key 29 (tests / A-B runs): 0 = the 64 x 64 workgroup everywhere
void conv_wgrad16_set_wide(int v) { g_wgrad16_wide = v; }
static int g_wgrad16_fold = 1;  // dsg_set_tuning key 39 (tests / A-B runs): 0 = Upsample2D's conv on the nine-tap kernel at full resolution
void conv_wgrad16_set_fold(int v) { g_wgrad16_fold = v; }
// Upsample2D's conv in the folded form (Wgrad16P.fold_co): the sampler form's shapes (wgrad16_ok) without GroupNorm in front
static bool wgrad16_fold(const dsg_conv_wgrad_args* a) {
  return g_wgrad16_fold && a->upsample == 1 && a->stride == 1 && a->ksize == 3 && a->c1 == 0 && a->gn_scale_shift == nullptr &&
         a->cout % 64 == 0 && a->win % 32 == 0 && a->hin % 2 == 0;
}
// co tiles per wave: 2 (a 64 ci x 128 co workgroup, one per CU) for the 3x3 gradients whose cout allows it
static int wgrad16_cot(int cout, int ksize) { return (g_wgrad16_wide && ksize == 3 && cout % 128 == 0) ? 2 : 1; }

static void wgrad16_runs(int cin, int cout, int n, int hout, int wout, int cot, int* strips, int* rsplit, int* spw = nullptr) {
  const int pairs = (cin / 64) * (cout / (64 * cot));
  *strips = n * (wout / 32);
  const int stages = hout / 2;
  if (cot == 2) {
    // One workgroup per CU: a grid of 1.5 x 256 workgroups runs as two rounds, the second half empty (measured: the 384-channel
    // layers 10 % SLOWER than with 64 x 64 workgroups, which share a CU and finish a ragged tail faster).  Among the splits
    // (several strips per workgroup, or several row ranges per strip) take the cheapest by a small model, in units of one
    // stage (~2.6 us): rounds x (stages per workgroup + 6 for its prologue and the 18-tile slab write) + 0.012 per workgroup
    // for its 295-KB partial slab going out and coming back through the reduce pass (calibrated on 384 -> 256 @ 64^2, B=32: 768
    // workgroups in 301 us -- the slabs mostly live in the Infinity Cache); ties go to the smaller grid.
    // (A pure "fill the rounds" rule took 3584 two-stage workgroups for 256 -> 256 @ 64^2 at batch 14.)
    int best_d = 1, best_r = 1, best_wg = 0;
    double best = 1e30;
    auto offer = [&](int d, int r) {
      const int wg = pairs * (*strips / d) * r;
      const double per_wg = (double)d * (stages / r);
      const double cost = cdiv(wg, 256) * (per_wg + 6.0) + 0.012 * wg;
      if (cost < best - 1e-9 || (cost < best + 1e-9 && wg < best_wg)) {
        best = cost; best_wg = wg; best_d = d; best_r = r;
      }
    };
    for (int d = 1; d <= *strips; ++d)
      if (*strips % d == 0) offer(d, 1);
    for (int r = 2; r <= stages; ++r)
      if (stages % r == 0) offer(1, r);
    *rsplit = best_r;
    if (spw) *spw = best_d;
    return;
  }
  const int want = std::max(1, cdiv(cot == 2 ? 256 : 512, pairs));  // workgroups wanted per (ci, co) block pair (2 per CU in all; 1 at cot 2)
  *rsplit = std::max(1, std::min(stages, cdiv(want, *strips)));
  int per = 1;  // strips per workgroup: the largest divisor of `strips` that still leaves `want` workgroups per pair
  if (*rsplit == 1)
    for (int d = 1; d <= *strips / want; ++d)
      if (*strips % d == 0) per = d;
  if (spw) *spw = per;
}

// (hout, wout): the kernel's K grid -- the conv's output map; for the two sampler forms the FULL-resolution one: the output map of
// Upsample2D's conv (2 hin x 2 win), the input map of Downsample2D's stride-2 conv (hin x win): wgrad16_kgrid
static void wgrad16_kgrid(const dsg_conv_wgrad_args* a, int* kh, int* kw) {
  *kh = a->upsample ? 2 * a->hin : a->hin;
  *kw = a->upsample ? 2 * a->win : a->win;
}
static bool wgrad16_ok(const dsg_conv_wgrad_args* a, int hout, int wout) {
  const int cin = a->c0 + a->c1, ctot = a->dy_ctotal ? a->dy_ctotal : a->cout;
  const bool sampler = (a->upsample != 0) != (a->stride == 2);   // one half-resolution operand (never both)
  const bool chans = (a->stride == 1 || a->stride == 2) && !(a->upsample && a->stride == 2) && cin % 64 == 0 &&
                     (a->c1 == 0 || a->c0 % 64 == 0) && a->cout % 64 == 0 && ctot % 8 == 0 && a->dy_coff % 64 == 0;
  if (sampler) return chans && a->ksize == 3 && a->c1 == 0 && wout % 64 == 0 && hout % 4 == 0;
  if (a->ksize == 1) return chans && (hout * wout) % 64 == 0;  // (re-tiled as rows of 32 pixels)
  return chans && a->ksize == 3 && wout % 32 == 0 && hout % 2 == 0;
}

static size_t wgrad16_ws_bytes(int cin, int cout, int ksize, int n, int hout, int wout) {
  if (ksize == 1) {
    hout = hout * wout / 32;
    wout = 32;
  }
  size_t most = 0;  // (of the two workgroup shapes: the tuning key may change between the query and the launch)
  if (ksize == 1) {  // (hout, wout: already re-tiled as rows of 32 pixels)
    int mi, nj, rpi;
    wgrad16_pw_plan(cin, 0, cout, n, hout * wout, &mi, &nj, &rpi);   // (nslab does not depend on the split between c0 and c1 ...
    int mi1, nj1, rpi1;
    wgrad16_pw_plan(64, cin - 64 > 0 ? cin - 64 : 0, cout, n, hout * wout, &mi1, &nj1, &rpi1);  // ... beyond MI = 1 vs 2)
    most = (size_t)n * std::max(rpi, rpi1) * ((size_t)cin * cout + cout) * sizeof(float);
  }
  for (int cot = 1; cot <= (ksize == 3 && cout % 128 == 0 ? 2 : 1); ++cot) {
    int strips, rsplit, spw;
    wgrad16_runs(cin, cout, n, hout, wout, cot, &strips, &rsplit, &spw);
    most = std::max(most, ((size_t)(strips / spw) * rsplit * ksize * ksize * cin * cout + (size_t)strips * rsplit * cout) * sizeof(float));  // + the dY-sum rows
  }
  return most;
}

// ... of the folded up-sampler form (4 taps x [cin][4 cout] per slab on x's own map)
static size_t wgrad16_fold_ws_bytes(const dsg_conv_wgrad_args* a) {
  size_t most = 0;
  const int cin = a->c0 + a->c1, c4 = 4 * a->cout;
  for (int cot = 1; cot <= (a->cout % 128 == 0 ? 2 : 1); ++cot) {
    int strips, rsplit, spw;
    wgrad16_runs(cin, c4, a->n, a->hin, a->win, cot, &strips, &rsplit, &spw);
    most = std::max(most, ((size_t)(strips / spw) * rsplit * 4 * cin * c4 + (size_t)strips * rsplit * c4) * sizeof(float));
  }
  return most;
}

static int launch_wgrad16_pw(const dsg_conv_wgrad_args* a, int plane, hipStream_t st) {
  Wgrad16PwP p;
  p.src0 = a->src0; p.src1 = a->src1; p.c0 = a->c0; p.c1 = a->c1; p.cin = a->c0 + a->c1; p.n = a->n; p.plane = plane;
  p.cout = a->cout; p.dy = a->dy; p.dy_ctotal = a->dy_ctotal ? a->dy_ctotal : a->cout; p.dy_coff = a->dy_coff;
  p.ss = a->gn_scale_shift; p.silu = a->silu; p.ws = static_cast<float*>(a->workspace);
  int mi, nj, rpi;
  wgrad16_pw_plan(p.c0, p.c1, p.cout, p.n, plane, &mi, &nj, &rpi);
  p.ci_blocks = p.cin / (64 * mi);
  p.rpi = rpi;
  const int nslab = p.n * rpi;
  const size_t need = (size_t)nslab * ((size_t)p.cin * p.cout + p.cout) * sizeof(float);
  if (p.ws == nullptr || a->workspace_bytes < need)
    return fail(DSG_ERR_WORKSPACE_TOO_SMALL, "dsg_conv2d_wgrad: workspace %zu bytes < required %zu", a->workspace_bytes, need);
  p.dysum_ws = a->dy_sums ? p.ws + (size_t)nslab * p.cin * p.cout : nullptr;
  int pi = -1;
  if (prof_on())
    pi = prof_begin(29, 2.0 * p.n * plane * (double)p.cout * p.cin, 2.0 * ((double)p.n * p.cin * plane + (double)p.n * p.cout * plane), st);
  const dim3 grid(p.ci_blocks * (p.cout / (64 * nj)), nslab);
  const bool bf = a->compute_dtype == DSG_BF16;
  const int act = p.ss == nullptr ? 0 : (p.silu ? 1 : 2);
#define DSG_W16PW_L(PRC, ACTV, MIV, NJV)                                                                                     \
  hipLaunchKernelGGL((conv_wgrad16_pw_kernel<PRC, ACTV, MIV, NJV>), grid, dim3(256), (size_t)(2 * (2 * MIV + 2 * NJV) * 64 * 32 * 2), st, p)
#define DSG_W16PW_A(PRC, MIV, NJV)                                                                                           \
  do {                                                                                                                       \
    if (act == 0) DSG_W16PW_L(PRC, 0, MIV, NJV);                                                                             \
    else if (act == 1) DSG_W16PW_L(PRC, 1, MIV, NJV);                                                                        \
    else DSG_W16PW_L(PRC, 2, MIV, NJV);                                                                                      \
  } while (0)
#define DSG_W16PW_T(PRC)                                                                                                     \
  do {                                                                                                                       \
    if (mi == 2 && nj == 2) DSG_W16PW_A(PRC, 2, 2);                                                                          \
    else if (mi == 2) DSG_W16PW_A(PRC, 2, 1);                                                                                \
    else if (nj == 2) DSG_W16PW_A(PRC, 1, 2);                                                                                \
    else DSG_W16PW_A(PRC, 1, 1);                                                                                             \
  } while (0)
  if (bf) DSG_W16PW_T(1);
  else DSG_W16PW_T(2);
#undef DSG_W16PW_T
#undef DSG_W16PW_A
#undef DSG_W16PW_L
  DSG_LAUNCH_CHECK();
  const DysumJob job = dysum_job(a->dy_sums ? p.dysum_ws : nullptr, rpi, p.cout, p.n, a->dy_sums,
                                 a->dy_sums_stride ? a->dy_sums_stride : p.cout, a->dy_bias_grad);
  launch_wgrad_reduce(p.ws, nslab, 1, p.cin, p.cout, p.cin, p.cout, a->dw, st, &job);
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

static int launch_wgrad16(const dsg_conv_wgrad_args* a, int hout, int wout, hipStream_t st) {
  if (a->ksize == 1 && g_wgrad16_pw) return launch_wgrad16_pw(a, hout * wout, st);
  Wgrad16P p;
  const bool fold = wgrad16_fold(a);  // Upsample2D's conv: the K grid is x's own map, dY its space-to-depth image (Wgrad16P.fold_co)
  const int taps = fold ? 4 : a->ksize * a->ksize;
  if (a->ksize == 1) {  // pointwise: rows of 32 pixels
    hout = hout * wout / 32;
    wout = 32;
  }
  if (fold) {
    hout = a->hin;
    wout = a->win;
  }
  p.src0 = a->src0; p.src1 = a->src1; p.c0 = a->c0; p.c1 = a->c1; p.cin = a->c0 + a->c1;
  p.n = a->n; p.h = hout; p.w = wout; p.cout = fold ? 4 * a->cout : a->cout; p.fold_co = fold ? a->cout : 0;
  p.dy = a->dy; p.dy_ctotal = a->dy_ctotal ? a->dy_ctotal : a->cout; p.dy_coff = a->dy_coff;
  p.ss = a->gn_scale_shift; p.silu = a->silu; p.ws = static_cast<float*>(a->workspace);
  p.xsh = (a->upsample && !fold) ? 1 : 0;
  p.dsh = a->stride == 2 ? 1 : 0;
  p.tiles_x = wout / 32; p.stages = hout / 2; p.ci_blocks = p.cin / 64;
  int strips, rsplit, spw;
  const int cot = wgrad16_cot(a->cout, a->ksize);   // (by the conv's own cout: a folded co block stays inside one parity; 64-co
                                                    // workgroups for every folded layer measured +0.2 % on the bf16 step)
  wgrad16_runs(p.cin, p.cout, p.n, hout, wout, cot, &strips, &rsplit, &spw);
  p.nrs = rsplit;
  p.spw = spw;
  const int nslab = (strips / spw) * rsplit;
  const size_t need = ((size_t)nslab * taps * p.cin * p.cout + (size_t)strips * rsplit * p.cout) * sizeof(float);
  if (p.ws == nullptr || a->workspace_bytes < need)
    return fail(DSG_ERR_WORKSPACE_TOO_SMALL, "dsg_conv2d_wgrad: workspace %zu bytes < required %zu", a->workspace_bytes, need);
  p.dysum_ws = a->dy_sums ? p.ws + (size_t)nslab * taps * p.cin * p.cout : nullptr;
  int pi = -1;
  if (prof_on())
    pi = prof_begin(29, fold ? 2.0 * p.n * (4.0 * hout * wout) * (double)a->cout * p.cin * 9
                             : 2.0 * p.n * (hout >> p.dsh) * (wout >> p.dsh) * (double)p.cout * p.cin * taps,   // (the reference op's FLOPs)
                    2.0 * ((double)p.n * p.cin * (p.h >> p.xsh) * (p.w >> p.xsh) + (double)p.n * p.cout * (hout >> p.dsh) * (wout >> p.dsh)), st);
  const dim3 grid(p.ci_blocks * (p.cout / (64 * cot)), nslab);
  const bool bf = a->compute_dtype == DSG_BF16;
  const int act = p.ss == nullptr ? 0 : (p.silu ? 1 : 2);
#define DSG_W16_LAUNCH(PRC, KSZ)                                                                                            \
  do {                                                                                                                      \
    if (act == 0) hipLaunchKernelGGL((conv_wgrad16_kernel<PRC, KSZ, 0>), grid, dim3(256), (size_t)W16Geom<KSZ>::LDS_BYTES, st, p); \
    else if (act == 1) hipLaunchKernelGGL((conv_wgrad16_kernel<PRC, KSZ, 1>), grid, dim3(256), (size_t)W16Geom<KSZ>::LDS_BYTES, st, p); \
    else hipLaunchKernelGGL((conv_wgrad16_kernel<PRC, KSZ, 2>), grid, dim3(256), (size_t)W16Geom<KSZ>::LDS_BYTES, st, p);    \
  } while (0)
#define DSG_W16_LAUNCH_WIDE(PRC)                                                                                             \
  do {                                                                                                                      \
    constexpr size_t lds = (size_t)W16Geom<3, 2>::LDS_BYTES;                                                                \
    if (act == 0) hipLaunchKernelGGL((conv_wgrad16_kernel<PRC, 3, 0, 2>), grid, dim3(256), lds, st, p);                     \
    else if (act == 1) hipLaunchKernelGGL((conv_wgrad16_kernel<PRC, 3, 1, 2>), grid, dim3(256), lds, st, p);                \
    else hipLaunchKernelGGL((conv_wgrad16_kernel<PRC, 3, 2, 2>), grid, dim3(256), lds, st, p);                              \
  } while (0)
  if (fold) {
    constexpr size_t lds1 = (size_t)W16Geom<3, 1>::LDS_BYTES, lds2 = (size_t)W16Geom<3, 2>::LDS_BYTES;
    if (cot == 2) {
      if (bf) hipLaunchKernelGGL((conv_wgrad16_kernel<1, 3, 0, 2, 1>), grid, dim3(256), lds2, st, p);
      else hipLaunchKernelGGL((conv_wgrad16_kernel<2, 3, 0, 2, 1>), grid, dim3(256), lds2, st, p);
    } else {
      if (bf) hipLaunchKernelGGL((conv_wgrad16_kernel<1, 3, 0, 1, 1>), grid, dim3(256), lds1, st, p);
      else hipLaunchKernelGGL((conv_wgrad16_kernel<2, 3, 0, 1, 1>), grid, dim3(256), lds1, st, p);
    }
  } else if (cot == 2) {
    if (bf) DSG_W16_LAUNCH_WIDE(1);
    else DSG_W16_LAUNCH_WIDE(2);
  } else if (a->ksize == 3) {
    if (bf) DSG_W16_LAUNCH(1, 3);
    else DSG_W16_LAUNCH(2, 3);
  } else {
    if (bf) DSG_W16_LAUNCH(1, 1);
    else DSG_W16_LAUNCH(2, 1);
  }
#undef DSG_W16_LAUNCH
#undef DSG_W16_LAUNCH_WIDE
  DSG_LAUNCH_CHECK();
  const int64_t slab = (int64_t)taps * p.cin * p.cout;
  (void)slab;
  DysumJob job = dysum_job(a->dy_sums ? p.dysum_ws : nullptr, p.tiles_x * rsplit, a->cout, p.n, a->dy_sums,
                           a->dy_sums_stride ? a->dy_sums_stride : a->cout, a->dy_bias_grad);
  if (fold) {
    job.parts = 4;
    const int main_blocks = (int)cdiv64((int64_t)p.cin * a->cout, 256);
    hipLaunchKernelGGL(wgrad_fold_reduce_kernel, dim3((unsigned)(main_blocks + job.blocks)), dim3(256), 0, st, p.ws, nslab, p.cin,
                       a->cout, a->dw, job, main_blocks);
  } else {
    launch_wgrad_reduce(p.ws, nslab, taps, p.cin, p.cout, p.cin, p.cout, a->dw, st, &job);
  }
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
  size_t need = (size_t)nslab * ks * ks * ci_blocks * cib * co_blocks * WG_CO * sizeof(float);
  if (ks == 3 && stride == 1 && cin % 32 == 0 && cout % 64 == 0) {  // the fp16x2-split kernel: one slab per run
    int strips, rsplit;
    wgrad_h2_runs(cin, cout, n, hout, wout, &strips, &rsplit);
    need = std::max(need, (size_t)strips * rsplit * (9 * (size_t)cin * cout + cout) * sizeof(float));  // + the dY-sum rows
  }
  if (wgrad_h2_pw_eligible(cin, cin, 0, cout, ks, stride, 0, hout * wout)) {
    int big, runs;
    wgrad_h2_pw_plan(cin, cout, n, hout * wout, &big, &runs);
    need = std::max(need, (size_t)runs * cin * cout * sizeof(float));
  }
  return need;
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
  DSG_CHECK_ARG(a->compute_dtype >= DSG_F32 && a->compute_dtype <= DSG_F16, "dsg_conv2d_wgrad: bad compute_dtype %d", a->compute_dtype);
  DSG_CHECK_ARG(a->dy_bias_grad == nullptr || a->dy_sums != nullptr, "dsg_conv2d_wgrad: dy_bias_grad rides on dy_sums (give both)");

  if (a->compute_dtype != DSG_F32) {  // mixed-precision tape: channel-blocked 16-bit x and dY
    int kh, kw;
    wgrad16_kgrid(a, &kh, &kw);
    DSG_CHECK_SHAPE(wgrad16_ok(a, kh, kw) && !a->force_direct,
                    "dsg_conv2d_wgrad: the 16-bit kernel takes 3x3 (stride 1, incl. the up-sampler's; stride 2) / 1x1 convs with "
                    "cin %% 64 == 0, cout %% 64 == 0 and (3x3) wout %% 32 == 0, hout %% 2 == 0 or (1x1) h * w %% 64 == 0 (got k %d, "
                    "stride %d, upsample %d, cin %d + %d, cout %d, %dx%d); convert to fp32 [N,C,H,W] for the rest", a->ksize,
                    a->stride, a->upsample, a->c0, a->c1, a->cout, a->hin, a->win);
    return launch_wgrad16(a, kh, kw, st);
  }
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
  p.dysum_ws = nullptr;
  const bool tile_ok = (p.wout % 32 == 0) && (p.hout % WG_SR == 0) && !a->force_direct;
  if (tile_ok) {
    const int k = a->ksize, s = a->stride, u = a->upsample;
    const bool small_ci = p.cin <= 32;
    if (wgrad_h2_eligible(p, k, s, u)) return launch_wgrad_h2(p, a->workspace_bytes, st, a->dy_sums, a->dy_sums_stride, a->dy_bias_grad);
    DSG_CHECK_ARG(a->dy_sums == nullptr,
                  "dsg_conv2d_wgrad: dy_sums is a by-product of the 16-bit kernel and of the fp32 split 3x3 kernel (stride 1, "
                  "cin %% 32 == 0, cout %% 64 == 0, wout %% 32 == 0, hout %% 2 == 0) only");
    if (k == 3 && s == 1 && u == 0) return small_ci ? launch_wgrad<3, 1, 0, 1>(p, a->workspace_bytes, st) : launch_wgrad<3, 1, 0, 2>(p, a->workspace_bytes, st);
    if (k == 3 && s == 1 && u == 1) return small_ci ? launch_wgrad<3, 1, 1, 1>(p, a->workspace_bytes, st) : launch_wgrad<3, 1, 1, 2>(p, a->workspace_bytes, st);
    if (k == 3 && s == 2) return launch_wgrad<3, 2, 0, 1>(p, a->workspace_bytes, st);
    if (wgrad_h2_pw_eligible(p.cin, p.c0, p.c1, p.cout, k, s, u, p.hin * p.win)) return launch_wgrad_h2_pw(p, a->workspace_bytes, st);
    if (k == 1 && s == 1 && u == 0) return small_ci ? launch_wgrad<1, 1, 0, 1>(p, a->workspace_bytes, st) : launch_wgrad<1, 1, 0, 2>(p, a->workspace_bytes, st);
  }
  DSG_CHECK_ARG(a->dy_sums == nullptr, "dsg_conv2d_wgrad: dy_sums given but the split kernels do not serve this call");
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
  if (a->compute_dtype != DSG_F32) {
    int kh, kw;
    dsg::wgrad16_kgrid(a, &kh, &kw);
    DSG_CHECK_SHAPE(dsg::wgrad16_ok(a, kh, kw), "dsg_conv2d_wgrad_workspace_bytes: shape not served by the 16-bit kernel");
    *bytes = dsg::wgrad16_ws_bytes(a->c0 + a->c1, a->cout, a->ksize, a->n, kh, kw);
    if (a->upsample == 1 && a->ksize == 3 && a->c1 == 0 && a->cout % 64 == 0 && a->win % 32 == 0 && a->hin % 2 == 0)
      *bytes = std::max(*bytes, dsg::wgrad16_fold_ws_bytes(a));  // (either form: the tuning key may change between the query and the launch)
    return DSG_OK;
  }
  *bytes = a->force_direct ? 0 : dsg::wgrad_ws_bytes(a->c0 + a->c1, a->cout, a->ksize, a->stride, hout, wout, a->n);
  return DSG_OK;
}
