// Probe (round 4): a synthetic NEIGHBOUR process for tools/probes/probe_lds_read2 (modes 13 - 15: packed fp32 VALU ops with op_sel that go
// wrong on lanes 48 - 63 next to some kernels of other processes).  Loops one kind of work forever:
//   0  v_mfma_f32_32x32x16_f16 with the accumulators in AGPRs            1  the same with the accumulators in VGPRs
//   2  v_mfma_f32_32x32x8_f16 (the pre-gfx950 shape), AGPR accumulators   3  plain VALU FMAs (control)
//   4  global_load_lds_dwordx4 (LDS DMA) + ds_read_b128                   5  mode 0 + mode 4 in one kernel
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_neighbour tools/probes/probe_neighbour.hip ;  run: ./probe_neighbour <mode>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void nb(float* out, const float* src, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  half8 fa, fb;
  half4 ga, gb;
  for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(0.001f * threadIdx.x); fb[j] = (_Float16)(0.002f * j); }
  for (int j = 0; j < 4; ++j) { ga[j] = fa[j]; gb[j] = fb[j]; }
  float v = threadIdx.x * 0.5f;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0 || MODE == 5)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\t"
                   "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_f16 %3, %4, %5, %3"
                   : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "v"(fa), "v"(fb));
    if constexpr (MODE == 1)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\t"
                   "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_f16 %3, %4, %5, %3"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fa), "v"(fb));
    if constexpr (MODE == 2)
      asm volatile("v_mfma_f32_32x32x8_f16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x8_f16 %1, %4, %5, %1\n\t"
                   "v_mfma_f32_32x32x8_f16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x8_f16 %3, %4, %5, %3"
                   : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "v"(ga), "v"(gb));
    if constexpr (MODE == 3) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v = fmaf(v, 0.999f, 0.001f);
    }
    if constexpr (MODE == 4 || MODE == 5) {
      const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds + (threadIdx.x >> 6) * 4096 + (it & 3) * 1024);
      const char* sp = reinterpret_cast<const char*>(src) + ((size_t)blockIdx.x * 4 + (it & 3)) * 4096;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n" ::"v"((threadIdx.x & 63) * 16), "s"(sp),
                   "s"(__builtin_amdgcn_readfirstlane(lbase)) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float4 r = *reinterpret_cast<const float4*>(lds + (threadIdx.x >> 6) * 4096 + (it & 3) * 1024 + (threadIdx.x & 63) * 16);
      q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w;
    }
  }
  float s = v + q.x + q.y + q.z + q.w;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  float *out, *src;
  hipMalloc(&out, 4096 * 256 * sizeof(float));
  hipMalloc(&src, (size_t)4096 * 4 * 4096);
  hipMemset(src, 0, (size_t)4096 * 4 * 4096);
  for (;;) {
    for (int k = 0; k < 20; ++k) {
      switch (mode) {
        case 0: hipLaunchKernelGGL(nb<0>, dim3(4096), dim3(256), 0, 0, out, src, 4000); break;
        case 1: hipLaunchKernelGGL(nb<1>, dim3(4096), dim3(256), 0, 0, out, src, 4000); break;
        case 2: hipLaunchKernelGGL(nb<2>, dim3(4096), dim3(256), 0, 0, out, src, 4000); break;
        case 3: hipLaunchKernelGGL(nb<3>, dim3(4096), dim3(256), 0, 0, out, src, 4000); break;
        case 4: hipLaunchKernelGGL(nb<4>, dim3(4096), dim3(256), 16384, 0, out, src, 4000); break;
        default: hipLaunchKernelGGL(nb<5>, dim3(4096), dim3(256), 16384, 0, out, src, 4000); break;
      }
    }
    hipDeviceSynchronize();
  }
  return 0;
}
