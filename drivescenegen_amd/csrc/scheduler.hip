// Noise-scheduler elementwise kernels for gfx950 (HBM-bound streaming, float4 per lane).
//
// Replaces the tensor arithmetic of diffusers' DDPMScheduler.add_noise / .step and
// DDIMScheduler.step (reference call sites: DriveSceneGen/pipeline/training_pipeline.py:80
// `noise_scheduler.add_noise`, and the DDPMPipeline loop behind training_pipeline.py:26-32 and
// DriveSceneGen/scripts/generation.py:14-20; formulas SURVEY.md App. A.3 / A.3b / A.4).
// Every expression is evaluated with individually rounded fp32 operations in the reference's
// order (fma contraction disabled for this file, IEEE division), so results are bit-identical to torch-CPU.
#include "dsg_common.h"

// HIP's __fmul_rn/__fadd_rn are plain operators (contractible); forbid fma contraction for this TU instead.
#pragma clang fp contract(off)

namespace dsg {

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// grid = (ceil(per_sample/1024), n)
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ nz,
                                                        const float* __restrict__ sa, const float* __restrict__ sb,
                                                        float* __restrict__ out, int64_t per) {
  const int n = blockIdx.y;
  const float a = sa[n], b = sb[n];
  const int64_t base = (int64_t)n * per;
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= per) return;
  if ((per & 3) == 0) {
    const float4 x = *reinterpret_cast<const float4*>(x0 + base + i0);
    const float4 e = *reinterpret_cast<const float4*>(nz + base + i0);
    float4 r;
    r.x = __fadd_rn(__fmul_rn(a, x.x), __fmul_rn(b, e.x));
    r.y = __fadd_rn(__fmul_rn(a, x.y), __fmul_rn(b, e.y));
    r.z = __fadd_rn(__fmul_rn(a, x.z), __fmul_rn(b, e.z));
    r.w = __fadd_rn(__fmul_rn(a, x.w), __fmul_rn(b, e.w));
    *reinterpret_cast<float4*>(out + base + i0) = r;
  } else {
    for (int k = 0; k < 4 && i0 + k < per; ++k)
      out[base + i0 + k] = __fadd_rn(__fmul_rn(a, x0[base + i0 + k]), __fmul_rn(b, nz[base + i0 + k]));
  }
}

__device__ __forceinline__ float pred_x0(float x, float e, float sb, float sa, float clip) {
  float v = __fdiv_rn(__fsub_rn(x, __fmul_rn(sb, e)), sa);
  if (clip > 0.f) v = clampf(v, -clip, clip);
  return v;
}

__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                        const float* __restrict__ nz, float* __restrict__ prev,
                                                        int64_t numel, float sb, float sa, float clip, float c0,
                                                        float ct, float sigma) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += stride) {
    const float xv = x[i];
    const float p0 = pred_x0(xv, eps[i], sb, sa, clip);
    float r = __fadd_rn(__fmul_rn(c0, p0), __fmul_rn(ct, xv));
    if (nz) r = __fadd_rn(r, __fmul_rn(sigma, nz[i]));
    prev[i] = r;
  }
}

__global__ __launch_bounds__(256) void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                        float* __restrict__ prev, int64_t numel, float sb, float sa,
                                                        float clip, float sap, float dc) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += stride) {
    const float e = eps[i];
    const float p0 = pred_x0(x[i], e, sb, sa, clip);
    prev[i] = __fadd_rn(__fmul_rn(sap, p0), __fmul_rn(dc, e));
  }
}

// (x/2 + 0.5).clamp(0,1), NCHW -> NHWC.  grid = (ceil(hw/256), n)
template <int MODE>
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ x, void* __restrict__ out, int c,
                                                          int hw) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  for (int ci = 0; ci < c; ++ci) {
    float v = x[((size_t)n * c + ci) * hw + p];
    v = clampf(__fadd_rn(__fdiv_rn(v, 2.0f), 0.5f), 0.f, 1.f);
    const size_t o = ((size_t)n * hw + p) * c + ci;
    if (MODE == 0) {
      reinterpret_cast<float*>(out)[o] = v;
    } else if (MODE == 1) {
      reinterpret_cast<uint8_t*>(out)[o] = (uint8_t)rintf(__fmul_rn(v, 255.0f));
    } else {
      reinterpret_cast<uint8_t*>(out)[o] = (uint8_t)__fmul_rn(v, 255.0f);
    }
  }
}

static inline int stream_blocks(int64_t numel) {
  int64_t b = cdiv64(numel, 256);
  return (int)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
}

}  // namespace dsg

DSG_API int dsg_add_noise(const float* x0, const float* noise, const float* sqrt_a, const float* sqrt_1ma, float* out,
                          int32_t n, int64_t per_sample, void* stream) {
  DSG_CHECK_ARG(x0 && noise && sqrt_a && sqrt_1ma && out, "dsg_add_noise: NULL pointer");
  DSG_CHECK_ARG(n > 0 && per_sample > 0 && n <= 65535, "dsg_add_noise: bad dims");
  hipLaunchKernelGGL(dsg::add_noise_kernel, dim3((unsigned)dsg::cdiv64(per_sample, 1024), n), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x0, noise, sqrt_a, sqrt_1ma, out, per_sample);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_ddpm_step(const float* sample, const float* eps, const float* noise, float* prev, int64_t numel,
                          float sqrt_beta_prod_t, float sqrt_alpha_prod_t, float clip, float coef_x0, float coef_xt,
                          float sigma, void* stream) {
  DSG_CHECK_ARG(sample && eps && prev, "dsg_ddpm_step: NULL pointer");
  DSG_CHECK_ARG(numel > 0, "dsg_ddpm_step: numel must be positive");
  hipLaunchKernelGGL(dsg::ddpm_step_kernel, dim3(dsg::stream_blocks(numel)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), sample, eps, noise, prev, numel, sqrt_beta_prod_t,
                     sqrt_alpha_prod_t, clip, coef_x0, coef_xt, sigma);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_ddim_step(const float* sample, const float* eps, float* prev, int64_t numel, float sqrt_beta_prod_t,
                          float sqrt_alpha_prod_t, float clip, float sqrt_alpha_prev, float dir_coef, void* stream) {
  DSG_CHECK_ARG(sample && eps && prev, "dsg_ddim_step: NULL pointer");
  DSG_CHECK_ARG(numel > 0, "dsg_ddim_step: numel must be positive");
  hipLaunchKernelGGL(dsg::ddim_step_kernel, dim3(dsg::stream_blocks(numel)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), sample, eps, prev, numel, sqrt_beta_prod_t, sqrt_alpha_prod_t,
                     clip, sqrt_alpha_prev, dir_coef);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_postprocess(const float* x, void* out, int32_t n, int32_t c, int32_t hw, int32_t mode, void* stream) {
  DSG_CHECK_ARG(x && out, "dsg_postprocess: NULL pointer");
  DSG_CHECK_ARG(n > 0 && c > 0 && hw > 0 && n <= 65535, "dsg_postprocess: bad dims");
  DSG_CHECK_ARG(mode >= 0 && mode <= 2, "dsg_postprocess: mode must be 0, 1 or 2");
  dim3 grid(dsg::cdiv(hw, 256), n);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 0) hipLaunchKernelGGL(dsg::postprocess_kernel<0>, grid, dim3(256), 0, st, x, out, c, hw);
  else if (mode == 1) hipLaunchKernelGGL(dsg::postprocess_kernel<1>, grid, dim3(256), 0, st, x, out, c, hw);
  else hipLaunchKernelGGL(dsg::postprocess_kernel<2>, grid, dim3(256), 0, st, x, out, c, hw);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}
