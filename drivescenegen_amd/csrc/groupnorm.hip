// GroupNorm for gfx950: per-channel fp64 statistics -> per-(n, c) scale/shift -> apply(+SiLU).
//
// Replaces nn.GroupNorm(32, C, eps=1e-5) (+ F.silu) at ResnetBlock2D.norm1/norm2,
// Attention.group_norm and conv_norm_out of diffusers' UNet2DModel (reference: the model built at
// DriveSceneGen/scripts/train.py:39-57).  HBM-bound streaming kernels:
// one float4 read per element for the statistics; the apply pass is normally folded into the
// consuming convolution's staging (conv.hip) and only runs standalone for tests.
//
// Statistics are kept per CHANNEL (sum, sum of squares, fp64) so that a group straddling the
// [x || skip] concat of the up path (6 of the 12 up-resnet norm1's) needs no concat tensor.
#include "dsg_h16.h"
#include <algorithm>

namespace dsg {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// grid = (c0 + c1, n); block = 256
__global__ __launch_bounds__(256) void gn_channel_stats_kernel(const float* __restrict__ src0, int c0,
                                                               const float* __restrict__ src1, int c1, int hw,
                                                               double* __restrict__ stats) {
  const int c = blockIdx.x, n = blockIdx.y;
  const int ct = c0 + c1;
  const float* sp = (c < c0) ? src0 + ((size_t)n * c0 + c) * hw : src1 + ((size_t)n * c1 + (c - c0)) * hw;
  double s = 0.0, ss = 0.0;
  if ((hw & 3) == 0) {
    const float4* sp4 = reinterpret_cast<const float4*>(sp);
    const int n4 = hw >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const float4 v = sp4[i];
      // fp32 partial over 4 elements, fp64 across (exact enough: 4-term fp32 sums feed an fp64 tree)
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    for (int i = threadIdx.x; i < hw; i += 256) {
      const double v = sp[i];
      s += v;
      ss += v * v;
    }
  }
  __shared__ double red[2][4];
  s = wave_sum(s);
  ss = wave_sum(ss);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = s;
    red[1][wave] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = stats + ((size_t)n * ct + c) * 2;
    o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// The same statistics of a channel-blocked tensor [N][C/8][hw][8]: grid = (C/8, n, splits); a thread walks the
// pixels of its split and keeps the 8 channels of its block; one partial per split, [N][C][splits][2].
// (dt: dsg_dtype of the tensor -- 0 fp32, 1 bf16, 2 fp16)
__global__ __launch_bounds__(256) void gn_channel_stats_blk_kernel(const void* __restrict__ srcv, int c, int hw_total,
                                                                   double* __restrict__ stats, int dt) {
  const int cb = blockIdx.x, n = blockIdx.y, sp_i = blockIdx.z, splits = gridDim.z;
  const int hw = hw_total / splits;
  const size_t base = ((size_t)n * c + cb * 8) * hw_total + (size_t)sp_i * hw * 8;
  const float4* sp = reinterpret_cast<const float4*>(static_cast<const float*>(srcv) + base);
  const uint4* sp16 = reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(srcv) + base);
  double s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.0;
#pragma unroll 4   // (several pixels' loads in flight per thread; the sums stay in pixel order)
  for (int i = threadIdx.x; i < hw; i += 256) {
    float v[8];
    if (dt) {
      const uint4 q = sp16[i];
      v[0] = word_lo(q.x, dt); v[1] = word_hi(q.x, dt); v[2] = word_lo(q.y, dt); v[3] = word_hi(q.y, dt);
      v[4] = word_lo(q.z, dt); v[5] = word_hi(q.z, dt); v[6] = word_lo(q.w, dt); v[7] = word_hi(q.w, dt);
    } else {
      const float4 a = sp[2 * i], b = sp[2 * i + 1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s[j] += (double)v[j];
      ss[j] += (double)v[j] * v[j];
    }
  }
  __shared__ double red[16][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double a = wave_sum(s[j]), b = wave_sum(ss[j]);
    if (lane == 0) {
      red[2 * j][wave] = a;
      red[2 * j + 1][wave] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int j = threadIdx.x >> 1, which = threadIdx.x & 1;
    stats[(((size_t)n * c + cb * 8 + j) * splits + sp_i) * 2 + which] =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
  }
}

// one thread per (n, c)
__global__ void gn_finalize_kernel(const double* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, int n, int c, int groups, int hw, float eps,
                                   float* __restrict__ out, float* __restrict__ mr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * c) return;
  const int ni = i / c, ci = i - ni * c;
  const int cpg = c / groups;
  const int g0 = (ci / cpg) * cpg;
  double s = 0.0, ss = 0.0;
  for (int k = 0; k < cpg; ++k) {
    const double* p = stats + ((size_t)ni * c + g0 + k) * 2;
    s += p[0];
    ss += p[1];
  }
  const double cnt = (double)cpg * (double)hw;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  const float sc = rstd * gamma[ci];
  out[2 * (size_t)i] = sc;
  out[2 * (size_t)i + 1] = beta[ci] - meanf * sc;
  if (mr) {  // training keeps (mean, rstd) per (n, c) for the backward pass
    mr[2 * (size_t)i] = meanf;
    mr[2 * (size_t)i + 1] = rstd;
  }
}

// One workgroup of FP_T threads per (n, group): per-tile partial statistics of the group's channels (which may sit in
// either of two concatenated sources) summed in a fixed order -- thread-strided over tiles, a fixed butterfly per wave,
// the waves in order.  One wave per group: the shallow levels' 512 partials per group are read four strides at a time,
// so their loads are in flight together (two round trips instead of eight); a 256-thread version of this kernel
// measured 11.8 us per launch against 6.8 -- four waves to start and a barrier for 64 values at the deep levels.
constexpr int FP_T = 64;
__global__ __launch_bounds__(FP_T) void gn_finalize_parts_kernel(const double* __restrict__ st0, int c0, int t0,
                                                              const double* __restrict__ st1, int c1, int t1,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int groups, int hw,
                                                              float eps, float* __restrict__ out,
                                                              float* __restrict__ mr, unsigned* __restrict__ bound) {
  const int c = c0 + c1, cpg = c / groups;
  const int ni = blockIdx.x / groups, g = blockIdx.x - ni * groups;
  const int lane = threadIdx.x;  // (index within the workgroup)
  // A group's channels are consecutive, hence its partials are one contiguous run of (sum, sum of squares) pairs per
  // source: the 64 lanes stride over that run (the deep levels have 16-32 channels x 4 tiles per group -- a loop
  // over channels with 4 active lanes would be 32 dependent round trips).  Fixed order for a given shape.
  double s = 0.0, ss = 0.0;
  float sq_max = 0.f;  // largest per-tile sum of squares: its square root bounds every |value| of the group
  const int ch0 = g * cpg, ch1 = ch0 + cpg;
  // gamma / beta of this lane's first channel go out with the partials' loads, not in a round trip of their own behind the
  // reduction (a launch of this kernel is 5 us of dependent latencies, 46 times a step)
  float gam0 = 0.f, bet0 = 0.f;
  if (lane < cpg) {
    gam0 = gamma[ch0 + lane];
    bet0 = beta[ch0 + lane];
  }
  {
    const int a = min(ch0, c0), b = min(ch1, c0);  // the group's channels that live in source 0
    const double2* p = reinterpret_cast<const double2*>(st0 + ((size_t)ni * c0 + a) * t0 * 2);
    const int cnt = (b - a) * t0;
    for (int i = lane; i < cnt; i += 4 * FP_T) {
      double2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = i + u * FP_T < cnt ? p[i + u * FP_T] : make_double2(0.0, 0.0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s += v[u].x;
        ss += v[u].y;
        sq_max = fmaxf(sq_max, (float)v[u].y);
      }
    }
  }
  if (c1 > 0) {
    const int a = max(ch0, c0) - c0, b = max(ch1, c0) - c0;  // ... and in source 1
    const double2* p = reinterpret_cast<const double2*>(st1 + ((size_t)ni * c1 + a) * t1 * 2);
    const int cnt = (b - a) * t1;
    for (int i = lane; i < cnt; i += 4 * FP_T) {
      double2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = i + u * FP_T < cnt ? p[i + u * FP_T] : make_double2(0.0, 0.0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s += v[u].x;
        ss += v[u].y;
        sq_max = fmaxf(sq_max, (float)v[u].y);
      }
    }
  }
  if (bound != nullptr) {  // range guard of the split convs that read these tensors un-normalised (dsg_conv_args.src_bound)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq_max = fmaxf(sq_max, __shfl_xor(sq_max, m));
    if ((lane & 63) == 0) atomicMax(bound + ni, __float_as_uint(sqrtf(sq_max)));
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    s += __shfl_xor(s, m);
    ss += __shfl_xor(ss, m);
  }
  const double cnt = (double)cpg * (double)hw;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  for (int k = lane; k < cpg; k += FP_T) {
    const int ch = g * cpg + k;
    const float sc = rstd * (k == lane ? gam0 : gamma[ch]);
    out[2 * ((size_t)ni * c + ch)] = sc;
    out[2 * ((size_t)ni * c + ch) + 1] = (k == lane ? bet0 : beta[ch]) - meanf * sc;
    if (mr) {  // training keeps (mean, rstd) per (n, c) for the backward pass
      mr[2 * ((size_t)ni * c + ch)] = meanf;
      mr[2 * ((size_t)ni * c + ch) + 1] = rstd;
    }
  }
}

// grid = (ceil(hw/1024), c, n)
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ src, const float* __restrict__ ssb,
                                                       int silu, float* __restrict__ dst, int c, int hw) {
  const int ci = blockIdx.y, n = blockIdx.z;
  const size_t base = ((size_t)n * c + ci) * hw;
  const float sc = ssb[((size_t)n * c + ci) * 2], sh = ssb[((size_t)n * c + ci) * 2 + 1];
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if ((hw & 3) == 0) {
    if (i0 < hw) {
      float4 v = *reinterpret_cast<const float4*>(src + base + i0);
      v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
      if (silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
      *reinterpret_cast<float4*>(dst + base + i0) = v;
    }
  } else {
    for (int k = 0; k < 4 && i0 + k < hw; ++k) {
      float v = src[base + i0 + k] * sc + sh;
      dst[base + i0 + k] = silu ? silu_f(v) : v;
    }
  }
}

// gridDim.y blocks per image: max over (channel, tile) of the sum of squares -> bits of its square root, atomically maxed
// (a maximum does not depend on the order; one block per image walked 256 dependent rounds of loads at the 256^2 levels: 13 us)
__global__ __launch_bounds__(256) void range_bound_kernel(const double* __restrict__ stats, int per_image,
                                                          unsigned* __restrict__ bound) {
  const int n = blockIdx.x;
  const double2* p = reinterpret_cast<const double2*>(stats) + (size_t)n * per_image;
  float m = 0.f;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < per_image; i += 256 * gridDim.y) m = fmaxf(m, (float)p[i].y);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(bound + n, __float_as_uint(sqrtf(m)));
}

// out[0] = max(out[0], max |x|) (out zeroed by the caller; non-negative floats order like their bits)
__global__ __launch_bounds__(256) void abs_max_kernel(const float* __restrict__ x, int64_t numel, float* __restrict__ out) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
}

}  // namespace dsg

DSG_API int dsg_range_bound_from_stats(const double* stats, int32_t n, int32_t c, int32_t tiles, uint32_t* bound, void* stream) {
  DSG_CHECK_ARG(stats && bound && n > 0 && c > 0 && tiles > 0, "dsg_range_bound_from_stats: bad argument");
  const int per_image = c * tiles, slices = std::min(32, std::max(1, per_image / 2048));
  hipLaunchKernelGGL(dsg::range_bound_kernel, dim3(n, slices), dim3(256), 0, static_cast<hipStream_t>(stream), stats, per_image, bound);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_abs_max(const float* x, int64_t numel, float* out, void* stream) {
  DSG_CHECK_ARG(x && out && numel > 0, "dsg_abs_max: bad argument");
  DSG_HIP(dsg::zero_words(out, 1, static_cast<hipStream_t>(stream)));
  const int blocks = (int)std::min<int64_t>(dsg::cdiv64(numel, 256 * 16), 512);
  hipLaunchKernelGGL(dsg::abs_max_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, static_cast<hipStream_t>(stream), x, numel, out);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_gn_channel_stats(const float* src0, int32_t c0, const float* src1, int32_t c1, int32_t n, int32_t hw,
                                 double* chan_stats, void* stream) {
  DSG_CHECK_ARG(src0 && chan_stats, "dsg_gn_channel_stats: NULL pointer");
  DSG_CHECK_ARG(c0 > 0 && c1 >= 0 && n > 0 && hw > 0, "dsg_gn_channel_stats: bad dims");
  DSG_CHECK_ARG((c1 == 0) == (src1 == nullptr), "dsg_gn_channel_stats: src1/c1 mismatch");
  DSG_CHECK_ARG(n <= 65535, "dsg_gn_channel_stats: batch too large for one launch");
  hipLaunchKernelGGL(dsg::gn_channel_stats_kernel, dim3(c0 + c1, n), dim3(256), 0, static_cast<hipStream_t>(stream),
                     src0, c0, src1, c1, hw, chan_stats);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_gn_channel_stats_blocked(const float* src, int32_t c, int32_t n, int32_t hw, int32_t splits,
                                         double* chan_stats, void* stream) {
  return dsg_gn_channel_stats_blocked_dt(src, c, n, hw, splits, chan_stats, DSG_F32, stream);
}

DSG_API int dsg_gn_channel_stats_blocked_dt(const void* src, int32_t c, int32_t n, int32_t hw, int32_t splits,
                                            double* chan_stats, int32_t dtype, void* stream) {
  DSG_CHECK_ARG(src && chan_stats, "dsg_gn_channel_stats_blocked: NULL pointer");
  DSG_CHECK_ARG(dtype >= DSG_F32 && dtype <= DSG_F16, "dsg_gn_channel_stats_blocked: bad dtype %d", dtype);
  DSG_CHECK_ARG(c > 0 && c % 8 == 0 && n > 0 && hw > 0, "dsg_gn_channel_stats_blocked: bad dims (C %% 8 != 0?)");
  DSG_CHECK_ARG(splits >= 1 && splits <= 65535 && hw % splits == 0,
                "dsg_gn_channel_stats_blocked: splits must divide hw (%d, %d)", splits, hw);
  DSG_CHECK_ARG(n <= 65535, "dsg_gn_channel_stats_blocked: batch too large for one launch");
  hipLaunchKernelGGL(dsg::gn_channel_stats_blk_kernel, dim3(c / 8, n, splits), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, c, hw, chan_stats, dtype);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

static int gn_finalize_impl(const double* chan_stats, const float* gamma, const float* beta, int32_t n, int32_t c,
                            int32_t groups, int32_t hw, float eps, float* scale_shift, float* mean_rstd, void* stream) {
  DSG_CHECK_ARG(chan_stats && gamma && beta && scale_shift, "dsg_gn_finalize: NULL pointer");
  DSG_CHECK_ARG(n > 0 && c > 0 && groups > 0 && hw > 0, "dsg_gn_finalize: bad dims");
  DSG_CHECK_ARG(c % groups == 0, "dsg_gn_finalize: channels (%d) not divisible by groups (%d)", c, groups);
  hipLaunchKernelGGL(dsg::gn_finalize_kernel, dim3(dsg::cdiv(n * c, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), chan_stats, gamma, beta, n, c, groups, hw, eps, scale_shift,
                     mean_rstd);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_gn_finalize(const double* chan_stats, const float* gamma, const float* beta, int32_t n, int32_t c,
                            int32_t groups, int32_t hw, float eps, float* scale_shift, void* stream) {
  return gn_finalize_impl(chan_stats, gamma, beta, n, c, groups, hw, eps, scale_shift, nullptr, stream);
}

DSG_API int dsg_gn_finalize_train(const double* chan_stats, const float* gamma, const float* beta, int32_t n, int32_t c,
                                  int32_t groups, int32_t hw, float eps, float* scale_shift, float* mean_rstd,
                                  void* stream) {
  DSG_CHECK_ARG(mean_rstd != nullptr, "dsg_gn_finalize_train: mean_rstd is NULL");
  return gn_finalize_impl(chan_stats, gamma, beta, n, c, groups, hw, eps, scale_shift, mean_rstd, stream);
}

static int gn_finalize_parts_impl(const double* stats0, int32_t c0, int32_t tiles0, const double* stats1, int32_t c1,
                                  int32_t tiles1, const float* gamma, const float* beta, int32_t n, int32_t groups,
                                  int32_t hw, float eps, float* scale_shift, float* mean_rstd, void* stream,
                                  uint32_t* bound = nullptr) {
  DSG_CHECK_ARG(stats0 && gamma && beta && scale_shift, "dsg_gn_finalize_parts: NULL pointer");
  DSG_CHECK_ARG((c1 == 0) == (stats1 == nullptr), "dsg_gn_finalize_parts: stats1/c1 mismatch");
  DSG_CHECK_ARG(n > 0 && c0 > 0 && c1 >= 0 && groups > 0 && hw > 0 && tiles0 > 0 && (c1 == 0 || tiles1 > 0),
                "dsg_gn_finalize_parts: bad dims");
  DSG_CHECK_ARG((c0 + c1) % groups == 0, "dsg_gn_finalize_parts: channels (%d) not divisible by groups (%d)",
                c0 + c1, groups);
  hipLaunchKernelGGL(dsg::gn_finalize_parts_kernel, dim3(n * groups), dim3(dsg::FP_T), 0, static_cast<hipStream_t>(stream),
                     stats0, c0, tiles0, stats1, c1, tiles1, gamma, beta, groups, hw, eps, scale_shift, mean_rstd, bound);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_gn_finalize_parts(const double* stats0, int32_t c0, int32_t tiles0, const double* stats1, int32_t c1,
                                  int32_t tiles1, const float* gamma, const float* beta, int32_t n, int32_t groups,
                                  int32_t hw, float eps, float* scale_shift, void* stream) {
  return gn_finalize_parts_impl(stats0, c0, tiles0, stats1, c1, tiles1, gamma, beta, n, groups, hw, eps, scale_shift,
                                nullptr, stream);
}

DSG_API int dsg_gn_finalize_parts_bound(const double* stats0, int32_t c0, int32_t tiles0, const double* stats1, int32_t c1,
                                        int32_t tiles1, const float* gamma, const float* beta, int32_t n, int32_t groups,
                                        int32_t hw, float eps, float* scale_shift, uint32_t* bound, void* stream) {
  return gn_finalize_parts_impl(stats0, c0, tiles0, stats1, c1, tiles1, gamma, beta, n, groups, hw, eps, scale_shift,
                                nullptr, stream, bound);
}

DSG_API int dsg_gn_finalize_parts_train(const double* stats0, int32_t c0, int32_t tiles0, const double* stats1,
                                        int32_t c1, int32_t tiles1, const float* gamma, const float* beta, int32_t n,
                                        int32_t groups, int32_t hw, float eps, float* scale_shift, float* mean_rstd,
                                        void* stream) {
  DSG_CHECK_ARG(mean_rstd != nullptr, "dsg_gn_finalize_parts_train: mean_rstd is NULL");
  return gn_finalize_parts_impl(stats0, c0, tiles0, stats1, c1, tiles1, gamma, beta, n, groups, hw, eps, scale_shift,
                                mean_rstd, stream);
}

DSG_API int dsg_gn_apply(const float* src, const float* scale_shift, int32_t silu, float* dst, int32_t n, int32_t c,
                         int32_t hw, void* stream) {
  DSG_CHECK_ARG(src && scale_shift && dst, "dsg_gn_apply: NULL pointer");
  DSG_CHECK_ARG(n > 0 && c > 0 && hw > 0 && c <= 65535 && n <= 65535, "dsg_gn_apply: bad dims");
  hipLaunchKernelGGL(dsg::gn_apply_kernel, dim3(dsg::cdiv(hw, 1024), c, n), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, scale_shift, silu, dst, c, hw);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}
