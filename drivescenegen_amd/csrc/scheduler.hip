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

// ---------------------------------------------------------------------------------------------------------------
// Counter-based device noise (opt-in replacement of the training loop's HOST draw, training_pipeline.py:72
// `torch.randn(batch.shape).to(device)`: 500 ms of one CPU thread for configs[4]'s [128, 8, 256, 256] against a 183-ms
// GPU step).  Philox4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
// Random123 known-answer vectors are in tests/test_oracle_kat.py) + Box-Muller.  THE STREAM IS DEFINED BY THIS TEXT and
// restated in oracle/philox_oracle.py:
//   element e of the flat tensor takes lane e % 4 of the block  Philox4x32-10(counter = (c.lo, c.hi, offset.lo, offset.hi),
//   key = (seed.lo, seed.hi)),  c = e / 4;  lanes (0, 1) and (2, 3) are Box-Muller pairs:
//     u1 = (float(r_even) + 0.5f) * 2^-32     in (0, 1]   (uint32 -> fp32 round-to-nearest-even; never 0: no log(0))
//     u2 =  float(r_odd)          * 2^-32     in [0, 1]
//     rad = sqrtf(-2 * logf(u1));   z_even = rad * cospif(2 * u2);   z_odd = rad * sinpif(2 * u2)
// A (seed, offset) pair names one tensor; the caller advances `offset` per draw (the training loop: its step counter, the
// rank in the high bits) -- no state lives on the device.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&r)[4]) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float& za, float& zb) {
  const float u1 = __fmul_rn(__fadd_rn((float)ra, 0.5f), 0x1p-32f);
  const float u2 = __fmul_rn((float)rb, 0x1p-32f);
  const float rad = sqrtf(__fmul_rn(-2.0f, logf(u1)));
  const float th = __fmul_rn(2.0f, u2);
  za = __fmul_rn(rad, cospif(th));
  zb = __fmul_rn(rad, sinpif(th));
}

// MODE 0: raw uint32 blocks (tests, and a general counter-based generator);  1: z ~ N(0,1) -> noise;
// 2: z -> noise AND noisy = sa[n]*x0 + sb[n]*z in the same pass (add_noise_kernel's two-multiply-one-add, bit for bit)
template <int MODE>
__global__ __launch_bounds__(256) void philox_kernel(const float* __restrict__ x0, const float* __restrict__ sa,
                                                     const float* __restrict__ sb, float* __restrict__ noisy,
                                                     void* __restrict__ noise_out, int64_t numel, int64_t per,
                                                     uint32_t seed_lo, uint32_t seed_hi, uint32_t off_lo, uint32_t off_hi) {
  const int64_t blocks = (numel + 3) >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const bool vec = (numel & 3) == 0 && (per & 3) == 0;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < blocks; c += stride) {
    uint32_t r[4];
    philox4x32_10((uint32_t)c, (uint32_t)((uint64_t)c >> 32), off_lo, off_hi, seed_lo, seed_hi, r);
    const int64_t e = c << 2;
    if (MODE == 0) {
      uint32_t* o = reinterpret_cast<uint32_t*>(noise_out);
      if ((numel & 3) == 0) {
        *reinterpret_cast<uint4*>(o + e) = make_uint4(r[0], r[1], r[2], r[3]);
      } else {
        for (int k = 0; k < 4 && e + k < numel; ++k) o[e + k] = r[k];
      }
      continue;
    }
    float z[4];
    box_muller(r[0], r[1], z[0], z[1]);
    box_muller(r[2], r[3], z[2], z[3]);
    float* o = reinterpret_cast<float*>(noise_out);
    if (vec) {
      *reinterpret_cast<float4*>(o + e) = make_float4(z[0], z[1], z[2], z[3]);
      if (MODE == 2) {
        const int64_t n = e / per;
        const float a = sa[n], b = sb[n];
        const float4 x = *reinterpret_cast<const float4*>(x0 + e);
        float4 q;
        q.x = __fadd_rn(__fmul_rn(a, x.x), __fmul_rn(b, z[0]));
        q.y = __fadd_rn(__fmul_rn(a, x.y), __fmul_rn(b, z[1]));
        q.z = __fadd_rn(__fmul_rn(a, x.z), __fmul_rn(b, z[2]));
        q.w = __fadd_rn(__fmul_rn(a, x.w), __fmul_rn(b, z[3]));
        *reinterpret_cast<float4*>(noisy + e) = q;
      }
    } else {
      for (int k = 0; k < 4 && e + k < numel; ++k) {
        o[e + k] = z[k];
        if (MODE == 2) {
          const int64_t n = (e + k) / per;
          noisy[e + k] = __fadd_rn(__fmul_rn(sa[n], x0[e + k]), __fmul_rn(sb[n], z[k]));
        }
      }
    }
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

namespace dsg {
static inline int philox_blocks(int64_t numel) {
  int64_t b = cdiv64(cdiv64(numel, 4), 256);
  return (int)(b < 1 ? 1 : (b > 256 * 32 ? 256 * 32 : b));
}
}  // namespace dsg

DSG_API int dsg_philox_u32(uint32_t* out, int64_t numel, uint64_t seed, uint64_t offset, void* stream) {
  DSG_CHECK_ARG(out, "dsg_philox_u32: NULL pointer");
  DSG_CHECK_ARG(numel > 0, "dsg_philox_u32: numel must be positive");
  hipLaunchKernelGGL(dsg::philox_kernel<0>, dim3(dsg::philox_blocks(numel)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (void*)out, numel,
                     (int64_t)4, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)offset, (uint32_t)(offset >> 32));
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_philox_normal(float* out, int64_t numel, uint64_t seed, uint64_t offset, void* stream) {
  DSG_CHECK_ARG(out, "dsg_philox_normal: NULL pointer");
  DSG_CHECK_ARG(numel > 0, "dsg_philox_normal: numel must be positive");
  hipLaunchKernelGGL(dsg::philox_kernel<1>, dim3(dsg::philox_blocks(numel)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (void*)out, numel,
                     (int64_t)4, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)offset, (uint32_t)(offset >> 32));
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_add_noise_philox(const float* x0, const float* sqrt_a, const float* sqrt_1ma, float* noisy, float* noise,
                                 int32_t n, int64_t per_sample, uint64_t seed, uint64_t offset, void* stream) {
  DSG_CHECK_ARG(x0 && sqrt_a && sqrt_1ma && noisy && noise, "dsg_add_noise_philox: NULL pointer");
  DSG_CHECK_ARG(n > 0 && per_sample > 0, "dsg_add_noise_philox: bad dims");
  const int64_t numel = (int64_t)n * per_sample;
  hipLaunchKernelGGL(dsg::philox_kernel<2>, dim3(dsg::philox_blocks(numel)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x0, sqrt_a, sqrt_1ma, noisy, (void*)noise, numel, per_sample, (uint32_t)seed, (uint32_t)(seed >> 32),
                     (uint32_t)offset, (uint32_t)(offset >> 32));
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

// The DEVICE address of a pinned host buffer (hipHostMalloc'ed or hipHostRegister'ed): what a kernel that reads the buffer in
// place must be given.  With torch's default pinned allocator the two addresses are equal (unified addressing); under its
// host-register configuration they need not be -- nothing here assumes it.  Fails (DSG_ERR_INVALID_ARG) for pageable memory.
DSG_API int dsg_host_device_pointer(const void* host, void** device) {
  DSG_CHECK_ARG(host && device, "dsg_host_device_pointer: NULL pointer");
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, const_cast<void*>(host), 0) != hipSuccess || d == nullptr) {
    (void)hipGetLastError();
    return dsg::fail(DSG_ERR_INVALID_ARG, "dsg_host_device_pointer: %p is not device-accessible pinned host memory", host);
  }
  *device = d;
  return DSG_OK;
}
