// Probe: issue model of v_mfma_f32_32x32x16_f16 on gfx950 with ONE wave per SIMD (the fp32-equivalent conv's operating
// point).  Cycles per MFMA for (a) independent accumulators, (b) back-to-back pairs on the same accumulator, (c) one
// accumulator throughout, and (d) independent accumulators with N plain / transcendental VALU instructions or an LDS
// read between consecutive MFMAs: how much other work hides behind one MFMA?
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_mfma tools/probes/probe_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define MFMA(ACC) "v_mfma_f32_32x32x16_f16 %" #ACC ", %8, %9, %" #ACC "\n\t"
#define FMA(R) "v_fma_f32 %" #R ", %" #R ", %14, %" #R "\n\t"
#define EXP(R) "v_exp_f32 %" #R ", %" #R "\n\t"
#define LDS(R) "ds_read_b128 %" #R ", %16\n\t"

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
  __syncthreads();
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {}, a5 = {}, a6 = {}, a7 = {};
  half8 fa, fb;
  for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(0.001f * threadIdx.x); fb[j] = (_Float16)(0.002f * j); }
  float v0 = 1.f, v1 = 2.f, v2 = 3.f, v3 = 4.f, c = 0.999f;
  f4 l0 = {};
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds + (threadIdx.x & 63) * 4);
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0)  // 8 independent accumulators, 3 rounds
      asm volatile(MFMA(0) MFMA(1) MFMA(2) MFMA(3) MFMA(4) MFMA(5) MFMA(6) MFMA(7)
                   MFMA(0) MFMA(1) MFMA(2) MFMA(3) MFMA(4) MFMA(5) MFMA(6) MFMA(7)
                   MFMA(0) MFMA(1) MFMA(2) MFMA(3) MFMA(4) MFMA(5) MFMA(6) MFMA(7)
                   : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+a"(a4), "+a"(a5), "+a"(a6), "+a"(a7)
                   : "v"(fa), "v"(fb), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(c), "v"(l0), "v"(laddr));
    else if constexpr (MODE == 1)  // the conv's order: one MFMA, then a dependent pair
      asm volatile(MFMA(0) MFMA(1) MFMA(1) MFMA(2) MFMA(3) MFMA(3) MFMA(4) MFMA(5) MFMA(5) MFMA(6) MFMA(7) MFMA(7)
                   MFMA(0) MFMA(1) MFMA(1) MFMA(2) MFMA(3) MFMA(3) MFMA(4) MFMA(5) MFMA(5) MFMA(6) MFMA(7) MFMA(7)
                   : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+a"(a4), "+a"(a5), "+a"(a6), "+a"(a7)
                   : "v"(fa), "v"(fb), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(c), "v"(l0), "v"(laddr));
    else if constexpr (MODE == 2)  // one accumulator
      asm volatile(MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0)
                   MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0)
                   MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0) MFMA(0)
                   : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+a"(a4), "+a"(a5), "+a"(a6), "+a"(a7)
                   : "v"(fa), "v"(fb), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(c), "v"(l0), "v"(laddr));
#define RND(X) MFMA(0) X MFMA(1) X MFMA(2) X MFMA(3) X MFMA(4) X MFMA(5) X MFMA(6) X MFMA(7) X
#define BODY(X)                                                                                         \
  asm volatile(RND(X) RND(X) RND(X)                                                                     \
               : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+a"(a4), "+a"(a5), "+a"(a6), "+a"(a7),        \
                 "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(l0)                                       \
               : "v"(fa), "v"(fb), "v"(c), "v"(laddr))
#undef MFMA
#undef FMA
#undef EXP
#undef LDS
#define MFMA(ACC) "v_mfma_f32_32x32x16_f16 %" #ACC ", %13, %14, %" #ACC "\n\t"
#define FMA(R) "v_fma_f32 %" #R ", %" #R ", %15, %" #R "\n\t"
#define EXP(R) "v_exp_f32 %" #R ", %" #R "\n\t"
#define LDS "ds_read_b128 %12, %16\n\t"
    else if constexpr (MODE == 3) BODY(FMA(8) FMA(9));
    else if constexpr (MODE == 4) BODY(FMA(8) FMA(9) FMA(10) FMA(11));
    else if constexpr (MODE == 5) BODY(FMA(8) FMA(9) FMA(10) FMA(11) FMA(8) FMA(9));
    else if constexpr (MODE == 6) BODY(FMA(8) FMA(9) FMA(10) FMA(11) FMA(8) FMA(9) FMA(10) FMA(11));
    else if constexpr (MODE == 7) BODY(EXP(8));
    else if constexpr (MODE == 8) BODY(EXP(8) EXP(9));
    else if constexpr (MODE == 9) BODY(EXP(8) FMA(9) FMA(10) FMA(11));
    else if constexpr (MODE == 10) BODY(LDS);
    else if constexpr (MODE == 11) BODY(LDS "s_waitcnt lgkmcnt(0)\n\t");  // an LDS read consumed at once
    else if constexpr (MODE == 12) BODY(FMA(8) FMA(8));                      // dependent VALU pair
    else if constexpr (MODE == 13) BODY(FMA(8) FMA(8) FMA(8) FMA(8));        // dependent VALU chain of 4
    else if constexpr (MODE == 14) BODY(EXP(8) FMA(8) EXP(8) FMA(8));        // the SiLU chain: exp -> add -> rcp -> mul
    else if constexpr (MODE == 15) BODY(FMA(8) FMA(9) FMA(10) FMA(11) FMA(8) FMA(9) FMA(10) FMA(11) FMA(8) FMA(9) FMA(10) FMA(11));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = v0 + v1 + v2 + v3 + l0.x;
  for (int j = 0; j < 16; ++j) s += a0[j] + a1[j] + a2[j] + a3[j] + a4[j] + a5[j] + a6[j] + a7[j];
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  if (s == 1234.5678f) out[2] = 1;
}

template <int MODE>
void run(unsigned long long* d, const char* what) {
  const int iters = 2000;
  hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(256), 0, 0, d, 10);
  hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(256), 0, 0, d, iters);
  unsigned long long h[2];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  const double n = 24.0 * iters;
  printf("%-58s %7.1f cycles/MFMA  (%.2f ns; clock %.2f GHz)\n", what, h[0] / n, h[1] * 10.0 / n, h[0] / (h[1] * 10.0));
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  run<0>(d, "8 independent accumulators");
  run<1>(d, "conv order: single + dependent pair");
  run<2>(d, "one accumulator");
  run<3>(d, "+ 2 v_fma between MFMAs");
  run<4>(d, "+ 4 v_fma");
  run<5>(d, "+ 6 v_fma");
  run<6>(d, "+ 8 v_fma");
  run<15>(d, "+ 12 v_fma");
  run<7>(d, "+ 1 v_exp");
  run<8>(d, "+ 2 v_exp");
  run<9>(d, "+ 1 v_exp + 3 v_fma");
  run<10>(d, "+ 1 ds_read_b128 (not waited for)");
  run<11>(d, "+ 1 ds_read_b128 + s_waitcnt lgkmcnt(0)");
  run<12>(d, "+ 2 dependent v_fma");
  run<13>(d, "+ 4 dependent v_fma");
  run<14>(d, "+ exp,fma,exp,fma dependent chain");
  return 0;
}
