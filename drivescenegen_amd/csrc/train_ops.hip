// Backward / optimizer kernels of the DDPM training step for gfx950 (all HBM-bound streaming passes).
//
// Reference: DriveSceneGen/pipeline/training_pipeline.py:84-91 -- `F.mse_loss`, `accelerator.backward`,
// `accelerator.clip_grad_norm_(model.parameters(), 1.0)`, `optimizer.step()` with torch.optim.AdamW
// (DriveSceneGen/scripts/train.py:66); layer semantics SURVEY.md App. A.2 / A.6.
//
//  - GroupNorm(+SiLU) backward in two streaming passes over (x, dy): per-(n, c) sums -> per-group
//    coefficients -> dx (the [x || skip] concat is split on the fly; an optional addend carries the
//    residual-path gradient so no separate add pass exists);
//  - per-(n, c) sums of a gradient tensor (bias and time-embedding gradients);
//  - small linear-layer backward (time MLP, time_emb_proj), SiLU backward;
//  - MSE loss forward+backward, global grad-norm (sum of squares), fused AdamW with the clip factor
//    applied on the fly.
#include "dsg_h16.h"
#include <cmath>

namespace dsg {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// d/du (u * sigmoid(u)); hardware exp / reciprocal (1 ulp each: the gradients' tolerances are 1e-4 and up) -- the IEEE
// division and the full-range expf were a third of the GroupNorm backward's instructions
__device__ __forceinline__ float dsilu(float u) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
  return s * (1.0f + u * (1.0f - s));
}

// block-wide sum of two doubles; result valid in thread 0
__device__ __forceinline__ void block_sum2(double& a, double& b) {
  __shared__ double red[2][4];
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = a;
    red[1][wave] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// grid = (c0 + c1, n).  s12[n][c] = (sum du, sum du * xhat), du = dy * silu'(u), u = x*sc + sh
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const float* __restrict__ src0, int c0,
                                                           const float* __restrict__ src1, int c1,
                                                           const float* __restrict__ dy, const float* __restrict__ ss,
                                                           const float* __restrict__ mr, int silu, int hw,
                                                           double* __restrict__ s12) {
  const int c = blockIdx.x, n = blockIdx.y, ct = c0 + c1;
  const float* xp = (c < c0) ? src0 + ((size_t)n * c0 + c) * hw : src1 + ((size_t)n * c1 + (c - c0)) * hw;
  const float* dp = dy + ((size_t)n * ct + c) * hw;
  const size_t k = ((size_t)n * ct + c) * 2;
  const float sc = ss[k], sh = ss[k + 1], mean = mr[k], rstd = mr[k + 1];
  double a = 0.0, b = 0.0;
  if ((hw & 3) == 0) {  // 16-byte loads; 4-term fp32 partials feed the fp64 sums
    const float4* xp4 = reinterpret_cast<const float4*>(xp);
    const float4* dp4 = reinterpret_cast<const float4*>(dp);
    auto accum = [&](const float4& x, float4 du) {
      if (silu) {
        du.x *= dsilu(x.x * sc + sh); du.y *= dsilu(x.y * sc + sh);
        du.z *= dsilu(x.z * sc + sh); du.w *= dsilu(x.w * sc + sh);
      }
      a += (double)((du.x + du.y) + (du.z + du.w));
      b += (double)((du.x * ((x.x - mean) * rstd) + du.y * ((x.y - mean) * rstd)) +
                    (du.z * ((x.z - mean) * rstd) + du.w * ((x.w - mean) * rstd)));
    };
    int i = threadIdx.x;
    for (; i + 256 < (hw >> 2); i += 512) {  // two 16-byte pairs in flight per thread (same order of additions)
      const float4 x0 = xp4[i], d0 = dp4[i], x1 = xp4[i + 256], d1 = dp4[i + 256];
      accum(x0, d0);
      accum(x1, d1);
    }
    if (i < (hw >> 2)) accum(xp4[i], dp4[i]);
  } else {
    for (int i = threadIdx.x; i < hw; i += 256) {
      const float x = xp[i];
      float du = dp[i];
      if (silu) du *= dsilu(x * sc + sh);
      a += du;
      b += (double)du * ((x - mean) * rstd);
    }
  }
  block_sum2(a, b);
  if (threadIdx.x == 0) {
    s12[k] = a;
    s12[k + 1] = b;
  }
}

// coef[n][c] = (rstd*gamma, rstd*g1/M, rstd*g2/M) and dgamma[c] += sum_n s2, dbeta[c] += sum_n s1 from the statistics pass's sums
// s12 = [N][C][splits][2] (split partials are added here, in split order).  A block owns 256 / n channels and ALL their images
// (thread = (channel, image)): the sums over the batch come out of LDS in image order -- the one-thread-per-(n, c) version had
// thread (0, c) walk the batch through global memory, 2 n dependent loads that were the kernel's whole 10.7 us, 46 times a step.
// (n <= 256; larger batches: cpb = 1 and a thread strides over the images)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const double* __restrict__ s12, const float* __restrict__ gamma,
                                                              const float* __restrict__ mr, int n, int c, int groups, int hw,
                                                              float* __restrict__ coef, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int splits, int cpb) {
  __shared__ double sh1[256], sh2[256];
  const int nn = n < 256 ? n : 256;          // threads per channel
  const int cl = threadIdx.x / nn, t = threadIdx.x - cl * nn;
  const int ci = blockIdx.x * cpb + cl;
  const bool live = cl < cpb && ci < c;
  const int cpg = c / groups;
  auto sum_of = [&](size_t nc, int which) -> double {
    double v = 0.0;
    for (int k = 0; k < splits; ++k) v += s12[(nc * splits + k) * 2 + which];
    return v;
  };
  double own1 = 0.0, own2 = 0.0;  // this thread's images' (s1, s2) of its channel, in image order
  if (live) {
    const int g0 = (ci / cpg) * cpg;
    for (int ni = t; ni < n; ni += nn) {
      double g1 = 0.0, g2 = 0.0;
#pragma unroll 4
      for (int k = 0; k < cpg; ++k) {
        const size_t nc = (size_t)ni * c + g0 + k;
        g1 += (double)gamma[g0 + k] * sum_of(nc, 0);
        g2 += (double)gamma[g0 + k] * sum_of(nc, 1);
      }
      const size_t i = (size_t)ni * c + ci;
      const double m = (double)cpg * (double)hw;
      const float rstd = mr[2 * i + 1];
      coef[3 * i] = rstd * gamma[ci];
      coef[3 * i + 1] = (float)(rstd * g1 / m);
      coef[3 * i + 2] = (float)(rstd * g2 / m);
      own1 += sum_of(i, 0);
      own2 += sum_of(i, 1);
    }
  }
  sh1[threadIdx.x] = own1;
  sh2[threadIdx.x] = own2;
  __syncthreads();
  if (live && t == 0) {
    double db = 0.0, dg = 0.0;
    for (int k = 0; k < nn; ++k) {
      db += sh1[cl * nn + k];
      dg += sh2[cl * nn + k];
    }
    dgamma[ci] += (float)dg;
    dbeta[ci] += (float)db;
  }
}

// The data-gradient conv's epilogue left per-tile partials (sum du, sum du * x) -- the RAW second moment -- in the forward
// statistics' table layout, parts[N][C][ntile][2] (dsg_conv_args.gnb_*).  One wave per (n, c): lane l adds tiles l, l + 64, ...
// in order, a fixed butterfly joins the lanes, and the pair becomes what the statistics pass would have written:
//   s12[n][c] = (S1, rstd * (S2raw - mean * S1))           (xhat = (x - mean) * rstd applied to the sums, in fp64)
__global__ __launch_bounds__(256) void gnb_parts_reduce_kernel(const double* __restrict__ parts, const float* __restrict__ mr,
                                                               int64_t nc_total, int ntile, double* __restrict__ s12) {
  const int lane = threadIdx.x & 63;
  const int64_t nc = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (nc >= nc_total) return;
  const double* pp = parts + nc * ntile * 2;
  double a = 0.0, b = 0.0;
  for (int t = lane; t < ntile; t += 64) {
    const double2 v = *reinterpret_cast<const double2*>(pp + 2 * t);
    a += v.x;
    b += v.y;
  }
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  if (lane == 0) {
    const double mean = (double)mr[2 * nc], rstd = (double)mr[2 * nc + 1];
    s12[2 * nc] = a;
    s12[2 * nc + 1] = rstd * (b - mean * a);
  }
}

// grid = (ceil(hw/(256*V)), c0+c1, n), V = 4 (16-byte accesses) when hw % 4 == 0 else 1.  dx = a*du - b - xhat*c2 (+ addend)
template <int V>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ src0, int c0,
                                                           const float* __restrict__ src1, int c1,
                                                           const float* __restrict__ dy, const float* __restrict__ ss,
                                                           const float* __restrict__ mr,
                                                           const float* __restrict__ coef, int silu, int hw,
                                                           const float* __restrict__ add0,
                                                           const float* __restrict__ add1, float* __restrict__ dx0,
                                                           float* __restrict__ dx1, const float* __restrict__ add0b) {
  const int c = blockIdx.y, n = blockIdx.z, ct = c0 + c1;
  const int i = (blockIdx.x * 256 + threadIdx.x) * V;
  if (i >= hw) return;
  const bool first = c < c0;
  const size_t xo = first ? ((size_t)n * c0 + c) * hw + i : ((size_t)n * c1 + (c - c0)) * hw + i;
  const float* xs = first ? src0 : src1;
  const float* as = first ? add0 : add1;
  const float* as2 = first ? add0b : nullptr;   // (a second fan-in term of source 0: skip-connection + residual gradients)
  float* ds = first ? dx0 : dx1;
  const size_t k = (size_t)n * ct + c;
  const float sc = ss[2 * k], sh = ss[2 * k + 1], mean = mr[2 * k], rstd = mr[2 * k + 1];
  const float k0 = coef[3 * k], k1 = coef[3 * k + 1], k2 = coef[3 * k + 2];
  float x[V], du[V], ad[V];
  if (V == 4) {
    const float4 x4 = *reinterpret_cast<const float4*>(xs + xo);
    const float4 d4 = *reinterpret_cast<const float4*>(dy + k * hw + i);
    x[0] = x4.x; x[1] = x4.y; x[2] = x4.z; x[3] = x4.w;
    du[0] = d4.x; du[1] = d4.y; du[2] = d4.z; du[3] = d4.w;
    if (as) {
      const float4 a4 = *reinterpret_cast<const float4*>(as + xo);
      ad[0] = a4.x; ad[1] = a4.y; ad[2] = a4.z; ad[3] = a4.w;
      if (as2) {   // (summed first, as the separate add pass did: the same bits)
        const float4 b4 = *reinterpret_cast<const float4*>(as2 + xo);
        ad[0] += b4.x; ad[1] += b4.y; ad[2] += b4.z; ad[3] += b4.w;
      }
    }
  } else {
    x[0] = xs[xo];
    du[0] = dy[k * hw + i];
    if (as) ad[0] = as[xo] + (as2 ? as2[xo] : 0.f);
  }
  float v[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    float d = du[e];
    if (silu) d *= dsilu(x[e] * sc + sh);
    v[e] = k0 * d - k1 - ((x[e] - mean) * rstd) * k2;
    if (as) v[e] += ad[e];
  }
  if (V == 4) *reinterpret_cast<float4*>(ds + xo) = make_float4(v[0], v[1], v[2], v[3]);
  else ds[xo] = v[0];
}

// grid = (c, n): out[n][c] = sum_hw x[n][c][:]
__global__ __launch_bounds__(256) void channel_sums_kernel(const float* __restrict__ x, int c, int hw,
                                                           float* __restrict__ out, int out_stride) {
  const int ci = blockIdx.x, n = blockIdx.y;
  const float* xp = x + ((size_t)n * c + ci) * hw;
  double a = 0.0, b = 0.0;
  if ((hw & 3) == 0) {
    const float4* xp4 = reinterpret_cast<const float4*>(xp);
    for (int i = threadIdx.x; i < (hw >> 2); i += 256) {
      const float4 v = xp4[i];
      a += (double)((v.x + v.y) + (v.z + v.w));
    }
  } else {
    for (int i = threadIdx.x; i < hw; i += 256) a += xp[i];
  }
  block_sum2(a, b);
  if (threadIdx.x == 0) out[(size_t)n * out_stride + ci] = (float)a;
}

// dst[c] += sum_n src[n*stride + c]
__global__ void reduce_rows_kernel(const float* __restrict__ src, int n, int c, int stride, float* __restrict__ dst) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= c) return;
  double s = 0.0;
#pragma unroll 8   // (loads in flight together; the sum stays in row order)
  for (int k = 0; k < n; ++k) s += src[(size_t)k * stride + ci];
  dst[ci] += (float)s;
}

// out = a + b (gradient fan-in of a tensor with several consumers)
__global__ __launch_bounds__(256) void add2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   int64_t numel, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256)
    out[i] = a[i] + b[i];
}

// dW[m][k] += sum_n dy[n][m] x[n][k]; one thread per (m, k)
__global__ void linear_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int n, int in_f, int out_f,
                                    int dy_stride, float* __restrict__ dw) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)in_f * out_f) return;
  const int k = (int)(i % in_f), m = (int)(i / in_f);
  float s = 0.f;
#pragma unroll 8   // (the batch's loads in flight together: 24 launches of this kernel per step are chains of round trips otherwise)
  for (int j = 0; j < n; ++j) s = fmaf(dy[(size_t)j * dy_stride + m], x[(size_t)j * in_f + k], s);
  dw[i] += s;
}

// dx[n][k] = sum_m dy[n][m] W[m][k]; a workgroup owns LD_NT images and 32 k's, its 8 wave-halves stride over m (the
// all-time-embedding projection has 5824 rows: one thread per (n, k) walked them serially in a millisecond) and are
// summed in a fixed order.  Several images per workgroup: W is 11.9 MB there and one image per workgroup pulled it through
// the L2s once per image (381 MB at batch 32, 153 us per call); per image the sum runs over m in the same order as before.
constexpr int LD_NT = 4;
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           int n, int in_f, int out_f, int dy_stride,
                                                           float* __restrict__ dx) {
  __shared__ float part[LD_NT][8][32];
  const int kt = blockIdx.x, j0 = blockIdx.y * LD_NT;
  const int kk = threadIdx.x & 31, ms = threadIdx.x >> 5;
  const int k = kt * 32 + kk;
  float s[LD_NT];
#pragma unroll
  for (int q = 0; q < LD_NT; ++q) s[q] = 0.f;
  if (k < in_f) {
#pragma unroll 4
    for (int m = ms; m < out_f; m += 8) {
      const float wv = w[(size_t)m * in_f + k];
#pragma unroll
      for (int q = 0; q < LD_NT; ++q)   // (images past the batch: row 0 again, never stored)
        s[q] = fmaf(dy[(size_t)(j0 + q < n ? j0 + q : 0) * dy_stride + m], wv, s[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < LD_NT; ++q) part[q][ms][kk] = s[q];
  __syncthreads();
  if (ms < LD_NT && k < in_f && j0 + ms < n) {  // wave-half q finishes image j0 + q
    float t = part[ms][0][kk];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += part[ms][i][kk];
    dx[(size_t)(j0 + ms) * in_f + k] = t;
  }
}

__global__ void silu_fwd_kernel(const float* __restrict__ z, int64_t numel, float* __restrict__ y) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < numel) y[i] = silu_f(z[i]);
}

// dz = dy * silu'(z)
__global__ void silu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dy, int64_t numel,
                                float* __restrict__ dz) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < numel) dz[i] = dy[i] * dsilu(z[i]);
}

// partial[b] = sum over the block's grid-stride slice of (a-b)^2 (or a^2 when b == nullptr); optionally writes
// dpred = coef * (a - b)
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             int64_t numel, float coef, float* __restrict__ dpred,
                                                             double* __restrict__ partial) {
  double s = 0.0, z = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
    const float d = b ? a[i] - b[i] : a[i];
    s += (double)d * d;
    if (dpred) dpred[i] = coef * d;
  }
  block_sum2(s, z);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// out[0] = scale * sum(partial) (mode 0) or sqrt(sum(partial)) (mode 1); fixed order -> deterministic
__global__ void finish_sum_kernel(const double* __restrict__ partial, int nb, double scale, int mode,
                                  float* __restrict__ out) {
  // one wave: lane t sums partials t, t + 64, ... in order, then a fixed butterfly (one thread walking all of them -- a few
  // thousand dependent adds behind as many loads -- was 135 us, twice a training step)
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  double s = 0.0;
#pragma unroll 8
  for (int i = threadIdx.x; i < nb; i += 64) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[0] = mode == 0 ? (float)(s * scale) : (float)sqrt(s);
}

// torch.optim.AdamW single-tensor semantics (decoupled weight decay), gradient scaled on the fly by
// min(1, max_norm / (total_norm + 1e-6)) (torch.nn.utils.clip_grad_norm_) when total_norm != nullptr.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t numel,
                                                    float lr, float beta1, float beta2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, const float* __restrict__ total_norm,
                                                    float max_norm) {
  float gs = 1.f;
  if (total_norm) {
    const float cf = max_norm / (total_norm[0] + 1e-6f);
    gs = cf < 1.f ? cf : 1.f;
  }
  const float step = lr / bc1;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gs;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);   // lerp form, as torch
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// in-place scale by min(1, max_norm/(norm+1e-6)) -- the standalone form of clip_grad_norm_
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, int64_t numel,
                                                         const float* __restrict__ total_norm, float max_norm) {
  const float cf = max_norm / (total_norm[0] + 1e-6f);
  if (cf >= 1.f) return;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) g[i] *= cf;
}

// GradScaler.unscale_: g *= inv in place; *found |= any(!isfinite(g)) (checked on the scaled-back value, like torch's
// _amp_foreach_non_finite_check_and_unscale_)
__global__ __launch_bounds__(256) void unscale_check_kernel(float* __restrict__ g, int64_t numel, float inv,
                                                            int* __restrict__ found) {
  bool bad = false;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
    const float v = g[i];
    bad = bad || !isfinite(v);
    g[i] = v * inv;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(found, 1);
}

// out = x * alpha[0] (alpha on the device: loss-scale / 1/world factors without a host sync)
__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, int64_t numel,
                                                    const float* __restrict__ alpha, float mult,
                                                    float* __restrict__ out) {
  const float a = (alpha ? alpha[0] : 1.f) * mult;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) out[i] = x[i] * a;
}

static inline int stream_blocks2(int64_t numel) {
  int64_t b = cdiv64(numel, 256 * 4);
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace dsg

using dsg::cdiv;

static int gn_bwd_f32_impl(const float* src0, int32_t c0, const float* src1, int32_t c1, const float* dy,
                           const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu, int32_t n,
                           int32_t hw, int32_t groups, const float* add0, const float* add1, float* dx0, float* dx1,
                           float* dgamma, float* dbeta, double* ws_s12, float* ws_coef, const double* parts, int32_t ntile,
                           void* stream, const float* add0b = nullptr) {
  DSG_CHECK_ARG(add0b == nullptr || add0 != nullptr, "dsg_gn_bwd: add0b without add0");
  DSG_CHECK_ARG(src0 && dy && scale_shift && mean_rstd && gamma && dx0 && dgamma && dbeta && ws_s12 && ws_coef,
                "dsg_gn_bwd: NULL pointer");
  DSG_CHECK_ARG(c0 > 0 && c1 >= 0 && n > 0 && hw > 0 && groups > 0, "dsg_gn_bwd: bad dims");
  DSG_CHECK_ARG((c1 == 0) == (src1 == nullptr) && (c1 == 0 || dx1 != nullptr), "dsg_gn_bwd: src1/dx1/c1 mismatch");
  const int c = c0 + c1;
  DSG_CHECK_ARG(c % groups == 0, "dsg_gn_bwd: channels not divisible by groups");
  DSG_CHECK_ARG(n <= 65535 && c <= 65535, "dsg_gn_bwd: grid too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (parts != nullptr)   // the data-gradient conv's epilogue already summed du and du * x per tile: no pass over x and dy
    hipLaunchKernelGGL(dsg::gnb_parts_reduce_kernel, dim3((unsigned)dsg::cdiv64((int64_t)n * c, 4)), dim3(256), 0, st, parts,
                       mean_rstd, (int64_t)n * c, ntile, ws_s12);
  else
    hipLaunchKernelGGL(dsg::gn_bwd_stats_kernel, dim3(c, n), dim3(256), 0, st, src0, c0, src1, c1, dy, scale_shift,
                       mean_rstd, silu, hw, ws_s12);
  DSG_LAUNCH_CHECK();
  {
    const int cpb = n < 256 ? 256 / n : 1;
    hipLaunchKernelGGL(dsg::gn_bwd_finalize_kernel, dim3(cdiv(c, cpb)), dim3(256), 0, st, ws_s12, gamma, mean_rstd, n, c, groups, hw,
                       ws_coef, dgamma, dbeta, 1, cpb);
  }
  DSG_LAUNCH_CHECK();
  if ((hw & 3) == 0)
    hipLaunchKernelGGL(dsg::gn_bwd_apply_kernel<4>, dim3(cdiv(hw, 1024), c, n), dim3(256), 0, st, src0, c0, src1, c1, dy,
                       scale_shift, mean_rstd, ws_coef, silu, hw, add0, add1, dx0, dx1, add0b);
  else
    hipLaunchKernelGGL(dsg::gn_bwd_apply_kernel<1>, dim3(cdiv(hw, 256), c, n), dim3(256), 0, st, src0, c0, src1, c1, dy,
                       scale_shift, mean_rstd, ws_coef, silu, hw, add0, add1, dx0, dx1, add0b);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_gn_bwd(const float* src0, int32_t c0, const float* src1, int32_t c1, const float* dy,
                       const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu, int32_t n,
                       int32_t hw, int32_t groups, const float* add0, const float* add1, float* dx0, float* dx1,
                       float* dgamma, float* dbeta, double* ws_s12, float* ws_coef, void* stream) {
  return gn_bwd_f32_impl(src0, c0, src1, c1, dy, scale_shift, mean_rstd, gamma, silu, n, hw, groups, add0, add1, dx0, dx1, dgamma,
                         dbeta, ws_s12, ws_coef, nullptr, 0, stream);
}

DSG_API int dsg_gn_bwd_add2(const float* src0, int32_t c0, const float* src1, int32_t c1, const float* dy,
                            const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu, int32_t n,
                            int32_t hw, int32_t groups, const float* add0, const float* add0b, const float* add1, float* dx0,
                            float* dx1, float* dgamma, float* dbeta, double* ws_s12, float* ws_coef, const double* parts,
                            int32_t ntile, void* stream) {
  DSG_CHECK_ARG((parts == nullptr) == (ntile == 0), "dsg_gn_bwd_add2: parts / ntile go together");
  return gn_bwd_f32_impl(src0, c0, src1, c1, dy, scale_shift, mean_rstd, gamma, silu, n, hw, groups, add0, add1, dx0, dx1, dgamma,
                         dbeta, ws_s12, ws_coef, parts, ntile, stream, add0b);
}

DSG_API int dsg_gn_bwd_parts(const float* src0, int32_t c0, const float* src1, int32_t c1, const float* dy,
                             const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu, int32_t n,
                             int32_t hw, int32_t groups, const float* add0, const float* add1, float* dx0, float* dx1,
                             float* dgamma, float* dbeta, double* ws_s12, float* ws_coef, const double* parts, int32_t ntile,
                             void* stream) {
  DSG_CHECK_ARG(parts != nullptr && ntile > 0, "dsg_gn_bwd_parts: parts / ntile missing");
  return gn_bwd_f32_impl(src0, c0, src1, c1, dy, scale_shift, mean_rstd, gamma, silu, n, hw, groups, add0, add1, dx0, dx1, dgamma,
                         dbeta, ws_s12, ws_coef, parts, ntile, stream);
}

DSG_API int dsg_channel_sums(const float* x, int32_t n, int32_t c, int32_t hw, float* out_nc, int32_t out_stride,
                             void* stream) {
  DSG_CHECK_ARG(x && out_nc, "dsg_channel_sums: NULL pointer");
  DSG_CHECK_ARG(n > 0 && c > 0 && hw > 0 && n <= 65535 && out_stride >= c, "dsg_channel_sums: bad dims");
  hipLaunchKernelGGL(dsg::channel_sums_kernel, dim3(c, n), dim3(256), 0, static_cast<hipStream_t>(stream), x, c, hw,
                     out_nc, out_stride);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_reduce_rows_add(const float* src, int32_t n, int32_t c, int32_t stride, float* dst, void* stream) {
  DSG_CHECK_ARG(src && dst, "dsg_reduce_rows_add: NULL pointer");
  DSG_CHECK_ARG(n > 0 && c > 0 && stride >= c, "dsg_reduce_rows_add: bad dims");
  hipLaunchKernelGGL(dsg::reduce_rows_kernel, dim3(cdiv(c, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), src, n,
                     c, stride, dst);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_linear_bwd(const float* x, const float* w, const float* dy, int32_t dy_stride, int32_t n, int32_t in_f,
                           int32_t out_f, float* dw, float* db, float* dx, void* stream) {
  DSG_CHECK_ARG(x && w && dy, "dsg_linear_bwd: NULL pointer");
  DSG_CHECK_ARG(n > 0 && in_f > 0 && out_f > 0 && dy_stride >= out_f, "dsg_linear_bwd: bad dims");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dw) {
    hipLaunchKernelGGL(dsg::linear_wgrad_kernel, dim3((unsigned)dsg::cdiv64((int64_t)in_f * out_f, 256)), dim3(256), 0,
                       st, x, dy, n, in_f, out_f, dy_stride, dw);
    DSG_LAUNCH_CHECK();
  }
  if (db) {
    hipLaunchKernelGGL(dsg::reduce_rows_kernel, dim3(cdiv(out_f, 256)), dim3(256), 0, st, dy, n, out_f, dy_stride, db);
    DSG_LAUNCH_CHECK();
  }
  if (dx) {
    hipLaunchKernelGGL(dsg::linear_dgrad_kernel, dim3(cdiv(in_f, 32), cdiv(n, dsg::LD_NT)), dim3(256), 0, st, dy, w, n, in_f, out_f,
                       dy_stride, dx);
    DSG_LAUNCH_CHECK();
  }
  return DSG_OK;
}

namespace dsg {
// dst[plane][2h][2w] = src[plane][h][w] (Upsample2D's nearest x2, materialised for the weight gradient of the
// up-sampler conv); one thread per source pixel pair -> two float4 stores
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ src, float* __restrict__ dst, int h,
                                                         int w, int64_t total2) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;  // index over (plane, y, x/2)
  if (i >= total2) return;
  const int w2 = w >> 1;
  const int x2 = (int)(i % w2);
  const int64_t r = i / w2;  // plane * h + y
  const float2 v = *reinterpret_cast<const float2*>(src + r * w + 2 * x2);
  const float4 o = make_float4(v.x, v.x, v.y, v.y);
  float* d = dst + (r * 2) * (2 * (int64_t)w) + 4 * x2;
  *reinterpret_cast<float4*>(d) = o;
  *reinterpret_cast<float4*>(d + 2 * w) = o;
}
// dst[plane][h][w] = sum of the 2x2 block of src[plane][2h][2w] (+ add): the adjoint of the nearest x2 upsample
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const float* __restrict__ src, const float* __restrict__ add,
                                                         float* __restrict__ dst, int h, int w, int64_t total2) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;  // index over (plane, y, x/2)
  if (i >= total2) return;
  const int w2 = w >> 1;
  const int x2 = (int)(i % w2);
  const int64_t r = i / w2;
  const float* s0 = src + (r * 2) * (2 * (int64_t)w) + 4 * x2;
  const float4 a = *reinterpret_cast<const float4*>(s0);
  const float4 b = *reinterpret_cast<const float4*>(s0 + 2 * w);
  float2 o = make_float2((a.x + a.y) + (b.x + b.y), (a.z + a.w) + (b.z + b.w));
  if (add) {
    const float2 e = *reinterpret_cast<const float2*>(add + r * w + 2 * x2);
    o.x += e.x;
    o.y += e.y;
  }
  *reinterpret_cast<float2*>(dst + r * w + 2 * x2) = o;
}

// ---------------------------------------------------------------------------------------------------------------
// The same passes for the mixed-precision training tape: channel-blocked 16-bit tensors [N][C/8][hw][8] (dt: 1 bf16,
// 2 fp16; include/dsg.h dsg_dtype).  A thread owns one pixel's 8 channels = one 16-byte access per tensor; all
// arithmetic in fp32, sums in fp64 across threads, results rounded once.
// ---------------------------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ void unpack8t(const uint4& q, float (&v)[8]) {
  v[0] = lo16<DT>(q.x); v[1] = hi16<DT>(q.x); v[2] = lo16<DT>(q.y); v[3] = hi16<DT>(q.y);
  v[4] = lo16<DT>(q.z); v[5] = hi16<DT>(q.z); v[6] = lo16<DT>(q.w); v[7] = hi16<DT>(q.w);
}
template <int DT>
__device__ __forceinline__ uint4 pack8t(const float (&v)[8]) {
  return make_uint4(pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3]), pack2<DT>(v[4], v[5]), pack2<DT>(v[6], v[7]));
}
__device__ __forceinline__ void unpack8(const uint4& q, int dt, float (&v)[8]) {
  v[0] = word_lo(q.x, dt); v[1] = word_hi(q.x, dt); v[2] = word_lo(q.y, dt); v[3] = word_hi(q.y, dt);
  v[4] = word_lo(q.z, dt); v[5] = word_hi(q.z, dt); v[6] = word_lo(q.w, dt); v[7] = word_hi(q.w, dt);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8], int dt) {
  return make_uint4(word_pack(v[0], v[1], dt), word_pack(v[2], v[3], dt), word_pack(v[4], v[5], dt), word_pack(v[6], v[7], dt));
}

// block-wide sums of 16 floats (8 channels x 2 quantities) -> doubles, fixed order; valid in threads 0..15
__device__ __forceinline__ double block_sum16(const float (&a)[8], const float (&b)[8]) {
  __shared__ double red16[16][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double x = wave_sum_d((double)a[j]), y = wave_sum_d((double)b[j]);
    if (lane == 0) {
      red16[2 * j][wave] = x;
      red16[2 * j + 1][wave] = y;
    }
  }
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x < 16) r = (red16[threadIdx.x][0] + red16[threadIdx.x][1]) + (red16[threadIdx.x][2] + red16[threadIdx.x][3]);
  return r;
}

// grid = ((c0 + c1) / 8, n, splits): partial[n][c][split] = (sum du, sum du * xhat) over the split's run of pixels
// (two pixels per thread and iteration, all four loads issued first: one 16-byte load pair per round trip ran at 2.8 TB/s)
template <int DT>
__global__ __launch_bounds__(256) void gn_bwd_stats_blk_kernel(const void* __restrict__ src0, int c0,
                                                               const void* __restrict__ src1, int c1,
                                                               const void* __restrict__ dy, const float* __restrict__ ss,
                                                               const float* __restrict__ mr, int silu, int hw_total,
                                                               double* __restrict__ part) {
  const int cb = blockIdx.x, n = blockIdx.y, sp = blockIdx.z, splits = gridDim.z, ct = c0 + c1;
  const int hw = hw_total / splits;
  const bool first = cb * 8 < c0;
  const unsigned short* xb = first ? static_cast<const unsigned short*>(src0) + ((size_t)n * c0 + cb * 8) * hw_total
                                   : static_cast<const unsigned short*>(src1) + ((size_t)n * c1 + (cb * 8 - c0)) * hw_total;
  const uint4* xp = reinterpret_cast<const uint4*>(xb) + (size_t)sp * hw;
  const uint4* dp = reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(dy) + ((size_t)n * ct + cb * 8) * hw_total) +
                    (size_t)sp * hw;
  const size_t k0 = ((size_t)n * ct + cb * 8) * 2;
  float sc[8], sh[8], mean[8], rstd[8], a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = ss[k0 + 2 * j]; sh[j] = ss[k0 + 2 * j + 1]; mean[j] = mr[k0 + 2 * j]; rstd[j] = mr[k0 + 2 * j + 1];
    a[j] = b[j] = 0.f;
  }
  auto accum = [&](const uint4& xq, const uint4& dq) {
    float x[8], du[8];
    unpack8t<DT>(xq, x);
    unpack8t<DT>(dq, du);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (silu) du[j] *= dsilu(x[j] * sc[j] + sh[j]);
      a[j] += du[j];
      b[j] += du[j] * ((x[j] - mean[j]) * rstd[j]);
    }
  };
  int i = threadIdx.x;
  for (; i + 256 < hw; i += 512) {  // (same per-thread order of additions as one pixel per iteration)
    const uint4 x0 = xp[i], d0 = dp[i], x1 = xp[i + 256], d1 = dp[i + 256];
    accum(x0, d0);
    accum(x1, d1);
  }
  if (i < hw) accum(xp[i], dp[i]);
  const double r = block_sum16(a, b);
  if (threadIdx.x < 16)
    part[(((size_t)n * ct + cb * 8 + (threadIdx.x >> 1)) * splits + sp) * 2 + (threadIdx.x & 1)] = r;
}

// s12[i] = sum over splits of part[i][split] (fixed order)
__global__ void sum_splits_kernel(const double* __restrict__ part, int64_t count, int splits, double* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // over (n, c, which)
  if (i >= count) return;
  const int64_t nc = i >> 1;
  const int which = (int)(i & 1);
  double s = 0.0;
  for (int k = 0; k < splits; ++k) s += part[(nc * splits + k) * 2 + which];
  out[i] = s;
}

// grid = (ceil(hw / 1024), (c0 + c1) / 8, n): dx = k0 * du - k1 - xhat * k2 (+ addend); a thread owns four pixels x 8
// channels (256 apart: coalesced), every load issued before the first use; the per-channel constants are block-uniform
template <int DT>
__global__ __launch_bounds__(256) void gn_bwd_apply_blk_kernel(const void* __restrict__ src0, int c0,
                                                               const void* __restrict__ src1, int c1,
                                                               const void* __restrict__ dy, const float* __restrict__ ss,
                                                               const float* __restrict__ mr, const float* __restrict__ coef,
                                                               int silu, int hw, const void* __restrict__ add0,
                                                               const void* __restrict__ add1, void* __restrict__ dx0,
                                                               void* __restrict__ dx1, const void* __restrict__ add0b) {
  constexpr int PX = 4;
  const int cb = blockIdx.y, n = blockIdx.z, ct = c0 + c1;
  const int i0 = blockIdx.x * (256 * PX) + threadIdx.x;
  const bool first = cb * 8 < c0;
  const size_t xbase = (first ? ((size_t)n * c0 + cb * 8) : ((size_t)n * c1 + (cb * 8 - c0))) * hw;  // elements
  const unsigned short* xs = static_cast<const unsigned short*>(first ? src0 : src1) + xbase;
  const unsigned short* as = static_cast<const unsigned short*>(first ? add0 : add1);
  const unsigned short* as2 = static_cast<const unsigned short*>(first ? add0b : nullptr);  // (a second fan-in term of source 0)
  unsigned short* ds = static_cast<unsigned short*>(first ? dx0 : dx1) + xbase;
  const size_t k = (size_t)n * ct + cb * 8;
  const unsigned short* dys = static_cast<const unsigned short*>(dy) + k * hw;
  float sc[8], sh[8], mean[8], rstd[8], q0[8], q1[8], q2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const size_t kj = k + j;
    sc[j] = ss[2 * kj]; sh[j] = ss[2 * kj + 1]; mean[j] = mr[2 * kj]; rstd[j] = mr[2 * kj + 1];
    q0[j] = coef[3 * kj]; q1[j] = coef[3 * kj + 1]; q2[j] = coef[3 * kj + 2];
  }
  uint4 xq[PX], dq[PX], aq[PX], bq[PX];
#pragma unroll
  for (int u = 0; u < PX; ++u) {
    const int i = min(i0 + 256 * u, hw - 1);  // (clamped: the tail's extra loads are not stored)
    xq[u] = *reinterpret_cast<const uint4*>(xs + (size_t)i * 8);
    dq[u] = *reinterpret_cast<const uint4*>(dys + (size_t)i * 8);
    if (as) aq[u] = *reinterpret_cast<const uint4*>(as + xbase + (size_t)i * 8);
    if (as2) bq[u] = *reinterpret_cast<const uint4*>(as2 + xbase + (size_t)i * 8);
  }
#pragma unroll
  for (int u = 0; u < PX; ++u) {
    float x[8], du[8], ad[8], ad2[8], v[8];
    unpack8t<DT>(xq[u], x);
    unpack8t<DT>(dq[u], du);
    if (as) unpack8t<DT>(aq[u], ad);
    if (as2) unpack8t<DT>(bq[u], ad2);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float d = du[j];
      if (silu) d *= dsilu(x[j] * sc[j] + sh[j]);
      v[j] = q0[j] * d - q1[j] - ((x[j] - mean[j]) * rstd[j]) * q2[j];
      if (as) v[j] += ad[j];
      if (as2) v[j] += ad2[j];
    }
    const int i = i0 + 256 * u;
    if (i < hw) *reinterpret_cast<uint4*>(ds + (size_t)i * 8) = pack8t<DT>(v);
  }
}

// grid = (c / 8, n): out[n][c] = sum over hw of a channel-blocked 16-bit tensor
__global__ __launch_bounds__(256) void channel_sums_blk_kernel(const void* __restrict__ x, int c, int hw, int dt,
                                                               float* __restrict__ out, int out_stride) {
  const int cb = blockIdx.x, n = blockIdx.y;
  const uint4* xp = reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(x) + ((size_t)n * c + cb * 8) * hw);
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = b[j] = 0.f;
  // two interleaved fp32 partial sums per channel, combined in fp64 (b carries the odd iterations)
  // (four loads per trip, issued before the first use: with few channels -- conv_out's 8 -- the grid is n workgroups, each
  //  a chain of hw / 256 round trips otherwise: 242 us for 32 MB; the sums keep their order: a, b, a, b)
  int i = threadIdx.x;
  for (; i + 768 < hw; i += 1024) {
    const uint4 q0 = xp[i], q1 = xp[i + 256], q2 = xp[i + 512], q3 = xp[i + 768];
    float v[8];
    unpack8(q0, dt, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += v[j];
    unpack8(q1, dt, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] += v[j];
    unpack8(q2, dt, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += v[j];
    unpack8(q3, dt, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] += v[j];
  }
  for (int it = 0; i < hw; i += 256, ++it) {
    float v[8];
    unpack8(xp[i], dt, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (it & 1) b[j] += v[j];
      else a[j] += v[j];
    }
  }
  const double r = block_sum16(a, b);
  __shared__ double pair[16];
  if (threadIdx.x < 16) pair[threadIdx.x] = r;
  __syncthreads();
  if (threadIdx.x < 8) out[(size_t)n * out_stride + cb * 8 + threadIdx.x] = (float)(pair[2 * threadIdx.x] + pair[2 * threadIdx.x + 1]);
}

// out = a + b over 16-bit values (8 per thread-step)
__global__ __launch_bounds__(256) void add2_16_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, int64_t n8,
                                                      int dt, uint4* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float x[8], y[8];
    unpack8(a[i], dt, x);
    unpack8(b[i], dt, y);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] += y[j];
    out[i] = pack8(x, dt);
  }
}

// channel-blocked [planes = N * C/8][h][w][8] -> [planes][2h][2w][8]: a pixel (16 bytes) is copied to its 2x2 block
__global__ __launch_bounds__(256) void upsample2x_blk_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int h,
                                                             int w, int64_t total) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;  // over (plane, y, x) of the source
  if (i >= total) return;
  const int x = (int)(i % w);
  const int64_t r = i / w;  // plane * h + y
  const uint4 v = src[i];
  uint4* d = dst + (r * 2) * (2 * (int64_t)w) + 2 * x;
  d[0] = v; d[1] = v; d[2 * w] = v; d[2 * w + 1] = v;
}

// the adjoint: [planes][2h][2w][8] -> [planes][h][w][8] sums of 2x2 blocks (+ add), fp32 arithmetic
__global__ __launch_bounds__(256) void sumpool2x2_blk_kernel(const uint4* __restrict__ src, const uint4* __restrict__ add,
                                                             uint4* __restrict__ dst, int h, int w, int dt, int64_t total) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;  // over (plane, y, x) of the result
  if (i >= total) return;
  const int x = (int)(i % w);
  const int64_t r = i / w;
  const uint4* s = src + (r * 2) * (2 * (int64_t)w) + 2 * x;
  float a[8], b[8], c[8], d[8];
  unpack8(s[0], dt, a); unpack8(s[1], dt, b); unpack8(s[2 * w], dt, c); unpack8(s[2 * w + 1], dt, d);
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = (a[j] + b[j]) + (c[j] + d[j]);
  if (add) {
    unpack8(add[i], dt, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += b[j];
  }
  dst[i] = pack8(a, dt);
}

// the same with the element type at compile time, two neighbouring results per thread (64 contiguous bytes per source row and
// thread, every load issued before the first use) and 32-bit index arithmetic (w even, total < 2^31): same sums in the same order
template <int DT>
__global__ __launch_bounds__(256) void sumpool2x2_blk2_kernel(const uint4* __restrict__ src, const uint4* __restrict__ add,
                                                              uint4* __restrict__ dst, int w, int pairs) {
  const int i2 = blockIdx.x * 256 + threadIdx.x;  // over (plane, y, x pair) of the result
  if (i2 >= pairs) return;
  const int wp = w >> 1;
  const int xp = i2 % wp, r = i2 / wp;             // r = plane * h + y
  const uint4* s = src + ((size_t)r * 2) * (2 * (size_t)w) + 4 * xp;
  uint4 q[8], aq[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    q[k] = s[k];
    q[4 + k] = s[2 * w + k];
  }
  const size_t o = (size_t)r * w + 2 * xp;
  if (add) {
    aq[0] = add[o];
    aq[1] = add[o + 1];
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float a[8], b[8], c[8], d[8];
    unpack8t<DT>(q[2 * e], a); unpack8t<DT>(q[2 * e + 1], b); unpack8t<DT>(q[4 + 2 * e], c); unpack8t<DT>(q[4 + 2 * e + 1], d);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (a[j] + b[j]) + (c[j] + d[j]);
    if (add) {
      unpack8t<DT>(aq[e], b);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += b[j];
    }
    dst[o + e] = pack8t<DT>(a);
  }
}
}  // namespace dsg

#define DSG_CHECK_DT16(dt, who) DSG_CHECK_ARG((dt) == DSG_BF16 || (dt) == DSG_F16, who ": dtype must be DSG_BF16 or DSG_F16")

DSG_API int dsg_gn_bwd_blocked(const void* src0, int32_t c0, const void* src1, int32_t c1, const void* dy,
                               const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu,
                               int32_t n, int32_t hw, int32_t groups, const void* add0, const void* add1, void* dx0,
                               void* dx1, float* dgamma, float* dbeta, double* ws_s12, float* ws_coef, int32_t dtype,
                               void* stream) {
  return dsg_gn_bwd_blocked_add2(src0, c0, src1, c1, dy, scale_shift, mean_rstd, gamma, silu, n, hw, groups, add0, nullptr, add1,
                                 dx0, dx1, dgamma, dbeta, ws_s12, ws_coef, dtype, stream);
}

static int gn_bwd_blocked_impl(const void* src0, int32_t c0, const void* src1, int32_t c1, const void* dy,
                               const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu,
                               int32_t n, int32_t hw, int32_t groups, const void* add0, const void* add0b, const void* add1,
                               void* dx0, void* dx1, float* dgamma, float* dbeta, double* ws_s12, float* ws_coef,
                               int32_t dtype, const double* parts, int32_t ntile, void* stream);

DSG_API int dsg_gn_bwd_blocked_add2(const void* src0, int32_t c0, const void* src1, int32_t c1, const void* dy,
                                    const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu,
                                    int32_t n, int32_t hw, int32_t groups, const void* add0, const void* add0b, const void* add1,
                                    void* dx0, void* dx1, float* dgamma, float* dbeta, double* ws_s12, float* ws_coef,
                                    int32_t dtype, void* stream) {
  return gn_bwd_blocked_impl(src0, c0, src1, c1, dy, scale_shift, mean_rstd, gamma, silu, n, hw, groups, add0, add0b, add1, dx0,
                             dx1, dgamma, dbeta, ws_s12, ws_coef, dtype, nullptr, 0, stream);
}

DSG_API int dsg_gn_bwd_blocked_parts(const void* src0, int32_t c0, const void* src1, int32_t c1, const void* dy,
                                     const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu,
                                     int32_t n, int32_t hw, int32_t groups, const void* add0, const void* add0b, const void* add1,
                                     void* dx0, void* dx1, float* dgamma, float* dbeta, double* ws_s12, float* ws_coef,
                                     int32_t dtype, const double* parts, int32_t ntile, void* stream) {
  DSG_CHECK_ARG(parts != nullptr && ntile > 0, "dsg_gn_bwd_blocked_parts: parts / ntile missing");
  return gn_bwd_blocked_impl(src0, c0, src1, c1, dy, scale_shift, mean_rstd, gamma, silu, n, hw, groups, add0, add0b, add1, dx0,
                             dx1, dgamma, dbeta, ws_s12, ws_coef, dtype, parts, ntile, stream);
}

static int gn_bwd_blocked_impl(const void* src0, int32_t c0, const void* src1, int32_t c1, const void* dy,
                               const float* scale_shift, const float* mean_rstd, const float* gamma, int32_t silu,
                               int32_t n, int32_t hw, int32_t groups, const void* add0, const void* add0b, const void* add1,
                               void* dx0, void* dx1, float* dgamma, float* dbeta, double* ws_s12, float* ws_coef,
                               int32_t dtype, const double* parts, int32_t ntile, void* stream) {
  using namespace dsg;
  DSG_CHECK_ARG(add0b == nullptr || add0 != nullptr, "dsg_gn_bwd_blocked_add2: add0b without add0");
  DSG_CHECK_ARG(src0 && dy && scale_shift && mean_rstd && gamma && dx0 && dgamma && dbeta && ws_s12 && ws_coef,
                "dsg_gn_bwd_blocked: NULL pointer");
  DSG_CHECK_DT16(dtype, "dsg_gn_bwd_blocked");
  DSG_CHECK_ARG(c0 > 0 && c1 >= 0 && c0 % 8 == 0 && c1 % 8 == 0 && n > 0 && hw > 0 && groups > 0, "dsg_gn_bwd_blocked: bad dims");
  DSG_CHECK_ARG((c1 == 0) == (src1 == nullptr) && (c1 == 0 || dx1 != nullptr), "dsg_gn_bwd_blocked: src1/dx1/c1 mismatch");
  const int c = c0 + c1;
  DSG_CHECK_ARG(c % groups == 0 && n <= 65535, "dsg_gn_bwd_blocked: channels not divisible by groups / batch too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int cpb = n < 256 ? 256 / n : 1;
  if (parts != nullptr) {  // the data-gradient conv's epilogue already summed du and du * x per tile: no pass over x and dy
    hipLaunchKernelGGL(gnb_parts_reduce_kernel, dim3((unsigned)cdiv64((int64_t)n * c, 4)), dim3(256), 0, st, parts, mean_rstd,
                       (int64_t)n * c, ntile, ws_s12);
    DSG_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(cdiv(c, cpb)), dim3(256), 0, st, ws_s12, gamma, mean_rstd, n, c, groups, hw, ws_coef,
                       dgamma, dbeta, 1, cpb);
    DSG_LAUNCH_CHECK();
    if (dtype == DSG_BF16)
      hipLaunchKernelGGL(gn_bwd_apply_blk_kernel<1>, dim3(cdiv(hw, 1024), c / 8, n), dim3(256), 0, st, src0, c0, src1, c1, dy,
                         scale_shift, mean_rstd, ws_coef, silu, hw, add0, add1, dx0, dx1, add0b);
    else
      hipLaunchKernelGGL(gn_bwd_apply_blk_kernel<2>, dim3(cdiv(hw, 1024), c / 8, n), dim3(256), 0, st, src0, c0, src1, c1, dy,
                         scale_shift, mean_rstd, ws_coef, silu, hw, add0, add1, dx0, dx1, add0b);
    DSG_LAUNCH_CHECK();
    return DSG_OK;
  }
  const int splits = dsg_gn_bwd_blocked_splits(hw);
  // ws_s12: [N][C][2] sums followed by the [N][C][splits][2] partials
  double* part = ws_s12 + (size_t)n * c * 2;
  if (dtype == DSG_BF16)
    hipLaunchKernelGGL(gn_bwd_stats_blk_kernel<1>, dim3(c / 8, n, splits), dim3(256), 0, st, src0, c0, src1, c1, dy, scale_shift,
                       mean_rstd, silu, hw, part);
  else
    hipLaunchKernelGGL(gn_bwd_stats_blk_kernel<2>, dim3(c / 8, n, splits), dim3(256), 0, st, src0, c0, src1, c1, dy, scale_shift,
                       mean_rstd, silu, hw, part);
  DSG_LAUNCH_CHECK();
  if (splits <= 2) {  // (the deep levels: the finalize pass adds the one or two partials itself, in the same order)
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(cdiv(c, cpb)), dim3(256), 0, st, part, gamma, mean_rstd, n, c, groups, hw, ws_coef,
                       dgamma, dbeta, splits, cpb);
  } else {
    hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)cdiv64((int64_t)n * c * 2, 256)), dim3(256), 0, st, part,
                       (int64_t)n * c * 2, splits, ws_s12);
    DSG_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(cdiv(c, cpb)), dim3(256), 0, st, ws_s12, gamma, mean_rstd, n, c, groups, hw, ws_coef,
                       dgamma, dbeta, 1, cpb);
  }
  DSG_LAUNCH_CHECK();
  if (dtype == DSG_BF16)
    hipLaunchKernelGGL(gn_bwd_apply_blk_kernel<1>, dim3(cdiv(hw, 1024), c / 8, n), dim3(256), 0, st, src0, c0, src1, c1, dy,
                       scale_shift, mean_rstd, ws_coef, silu, hw, add0, add1, dx0, dx1, add0b);
  else
    hipLaunchKernelGGL(gn_bwd_apply_blk_kernel<2>, dim3(cdiv(hw, 1024), c / 8, n), dim3(256), 0, st, src0, c0, src1, c1, dy,
                       scale_shift, mean_rstd, ws_coef, silu, hw, add0, add1, dx0, dx1, add0b);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_gn_bwd_blocked_splits(int32_t hw) {
  int splits = 1;
  while (splits < 16 && hw % (2 * splits) == 0 && hw / (2 * splits) >= 2048) splits *= 2;
  return splits;
}

DSG_API int dsg_channel_sums_blocked(const void* x, int32_t n, int32_t c, int32_t hw, float* out_nc, int32_t out_stride,
                                     int32_t dtype, void* stream) {
  DSG_CHECK_ARG(x && out_nc, "dsg_channel_sums_blocked: NULL pointer");
  DSG_CHECK_DT16(dtype, "dsg_channel_sums_blocked");
  DSG_CHECK_ARG(n > 0 && c > 0 && c % 8 == 0 && hw > 0 && n <= 65535 && out_stride >= c, "dsg_channel_sums_blocked: bad dims");
  hipLaunchKernelGGL(dsg::channel_sums_blk_kernel, dim3(c / 8, n), dim3(256), 0, static_cast<hipStream_t>(stream), x, c, hw,
                     dtype, out_nc, out_stride);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_add_dt(const void* a, const void* b, int64_t numel, void* out, int32_t dtype, void* stream) {
  if (dtype == DSG_F32)
    return dsg_add(static_cast<const float*>(a), static_cast<const float*>(b), numel, static_cast<float*>(out), stream);
  DSG_CHECK_DT16(dtype, "dsg_add_dt");
  DSG_CHECK_ARG(a && b && out && numel > 0 && numel % 8 == 0, "dsg_add_dt: bad argument (16-bit tensors: numel %% 8 == 0)");
  hipLaunchKernelGGL(dsg::add2_16_kernel, dim3(dsg::stream_blocks2(numel / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint4*>(a), static_cast<const uint4*>(b), numel / 8, dtype, static_cast<uint4*>(out));
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_upsample_nearest2x_blocked(const void* src, void* dst, int64_t planes, int32_t h, int32_t w, int32_t dtype,
                                           void* stream) {
  DSG_CHECK_ARG(src && dst && planes > 0 && h > 0 && w > 0, "dsg_upsample_nearest2x_blocked: bad argument");
  DSG_CHECK_DT16(dtype, "dsg_upsample_nearest2x_blocked");
  const int64_t total = planes * h * w;
  hipLaunchKernelGGL(dsg::upsample2x_blk_kernel, dim3((unsigned)dsg::cdiv64(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const uint4*>(src), static_cast<uint4*>(dst), h, w, total);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_sumpool2x2_blocked(const void* src, const void* add, void* dst, int64_t planes, int32_t h, int32_t w,
                                   int32_t dtype, void* stream) {
  DSG_CHECK_ARG(src && dst && planes > 0 && h > 0 && w > 0, "dsg_sumpool2x2_blocked: bad argument");
  DSG_CHECK_DT16(dtype, "dsg_sumpool2x2_blocked");
  const int64_t total = planes * h * w;
  if ((w & 1) == 0 && total < (int64_t)1 << 31) {  // (591 -> 283 us at 256 x 256 x 64 channels x 128 images, bf16: 2.7 -> 5.7 TB/s; bitwise the same)
    const int pairs = (int)(total / 2);
    if (dtype == DSG_BF16)
      hipLaunchKernelGGL(dsg::sumpool2x2_blk2_kernel<1>, dim3((unsigned)dsg::cdiv(pairs, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                         static_cast<const uint4*>(src), static_cast<const uint4*>(add), static_cast<uint4*>(dst), w, pairs);
    else
      hipLaunchKernelGGL(dsg::sumpool2x2_blk2_kernel<2>, dim3((unsigned)dsg::cdiv(pairs, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                         static_cast<const uint4*>(src), static_cast<const uint4*>(add), static_cast<uint4*>(dst), w, pairs);
    DSG_LAUNCH_CHECK();
    return DSG_OK;
  }
  hipLaunchKernelGGL(dsg::sumpool2x2_blk_kernel, dim3((unsigned)dsg::cdiv64(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const uint4*>(src), static_cast<const uint4*>(add),
                     static_cast<uint4*>(dst), h, w, dtype, total);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_upsample_nearest2x(const float* src, float* dst, int64_t planes, int32_t h, int32_t w, void* stream) {
  DSG_CHECK_ARG(src && dst && planes > 0 && h > 0 && w > 0 && (w & 1) == 0, "dsg_upsample_nearest2x: bad argument (w even)");
  const int64_t total2 = planes * h * (w / 2);
  hipLaunchKernelGGL(dsg::upsample2x_kernel, dim3((unsigned)dsg::cdiv64(total2, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, dst, h, w, total2);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_sumpool2x2(const float* src, const float* add, float* dst, int64_t planes, int32_t h, int32_t w,
                           void* stream) {
  DSG_CHECK_ARG(src && dst && planes > 0 && h > 0 && w > 0 && (w & 1) == 0, "dsg_sumpool2x2: bad argument (w even)");
  const int64_t total2 = planes * h * (w / 2);
  hipLaunchKernelGGL(dsg::sumpool2x2_kernel, dim3((unsigned)dsg::cdiv64(total2, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, add, dst, h, w, total2);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_add(const float* a, const float* b, int64_t numel, float* out, void* stream) {
  DSG_CHECK_ARG(a && b && out && numel > 0, "dsg_add: bad argument");
  hipLaunchKernelGGL(dsg::add2_kernel, dim3(dsg::stream_blocks2(numel)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     a, b, numel, out);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_scale(const float* x, int64_t numel, const float* alpha_dev, float mult, float* out, void* stream) {
  DSG_CHECK_ARG(x && out && numel > 0, "dsg_scale: bad argument");
  hipLaunchKernelGGL(dsg::scale_kernel, dim3(dsg::stream_blocks2(numel)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, numel, alpha_dev, mult, out);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_silu_fwd(const float* z, int64_t numel, float* y, void* stream) {
  DSG_CHECK_ARG(z && y && numel > 0, "dsg_silu_fwd: bad argument");
  hipLaunchKernelGGL(dsg::silu_fwd_kernel, dim3((unsigned)dsg::cdiv64(numel, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), z, numel, y);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_silu_bwd(const float* z, const float* dy, int64_t numel, float* dz, void* stream) {
  DSG_CHECK_ARG(z && dy && dz && numel > 0, "dsg_silu_bwd: bad argument");
  hipLaunchKernelGGL(dsg::silu_bwd_kernel, dim3((unsigned)dsg::cdiv64(numel, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), z, dy, numel, dz);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_mse_loss(const float* pred, const float* target, int64_t numel, float grad_scale, float* loss,
                         float* dpred, double* ws, size_t ws_bytes, void* stream) {
  DSG_CHECK_ARG(pred && target && loss && ws && numel > 0, "dsg_mse_loss: bad argument");
  const int nb = dsg::stream_blocks2(numel);
  if (ws_bytes < (size_t)nb * sizeof(double))
    return dsg::fail(DSG_ERR_WORKSPACE_TOO_SMALL, "dsg_mse_loss: workspace %zu < %zu", ws_bytes, (size_t)nb * 8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dsg::sqdiff_partial_kernel, dim3(nb), dim3(256), 0, st, pred, target, numel,
                     2.0f * grad_scale / (float)numel, dpred, ws);
  DSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(dsg::finish_sum_kernel, dim3(1), dim3(64), 0, st, ws, nb, 1.0 / (double)numel, 0, loss);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_l2_norm(const float* x, int64_t numel, float* norm, double* ws, size_t ws_bytes, void* stream) {
  DSG_CHECK_ARG(x && norm && ws && numel > 0, "dsg_l2_norm: bad argument");
  const int nb = dsg::stream_blocks2(numel);
  if (ws_bytes < (size_t)nb * sizeof(double))
    return dsg::fail(DSG_ERR_WORKSPACE_TOO_SMALL, "dsg_l2_norm: workspace %zu < %zu", ws_bytes, (size_t)nb * 8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dsg::sqdiff_partial_kernel, dim3(nb), dim3(256), 0, st, x, (const float*)nullptr, numel, 0.f,
                     (float*)nullptr, ws);
  DSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(dsg::finish_sum_kernel, dim3(1), dim3(64), 0, st, ws, nb, 1.0, 1, norm);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_clip_scale(float* g, int64_t numel, const float* total_norm, float max_norm, void* stream) {
  DSG_CHECK_ARG(g && total_norm && numel > 0 && max_norm > 0, "dsg_clip_scale: bad argument");
  hipLaunchKernelGGL(dsg::clip_scale_kernel, dim3(dsg::stream_blocks2(numel)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g, numel, total_norm, max_norm);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_unscale_check(float* g, int64_t numel, float inv_scale, int32_t* found_inf, void* stream) {
  DSG_CHECK_ARG(g && found_inf && numel > 0, "dsg_unscale_check: bad argument");
  hipLaunchKernelGGL(dsg::unscale_check_kernel, dim3(dsg::stream_blocks2(numel)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g, numel, inv_scale, found_inf);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, double lr,
                           double beta1, double beta2, double eps, double weight_decay, int64_t step,
                           const float* total_norm, float max_norm, void* stream) {
  DSG_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && numel > 0 && step >= 1, "dsg_adamw_step: bad argument");
  // hyper-parameters arrive as the Python doubles torch.optim.AdamW holds; bias corrections in fp64 like torch
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(dsg::adamw_kernel, dim3(dsg::stream_blocks2(numel)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), param, grad, exp_avg, exp_avg_sq, numel, (float)lr,
                     (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)bc1, (float)sqrt(bc2),
                     total_norm, max_norm);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}
