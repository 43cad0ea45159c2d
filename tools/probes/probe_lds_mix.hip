// Probe (round 4): which mix of LDS reads goes wrong next to another process?  conv_fewout_kernel's compute loop in isolation --
// per-lane patch reads + uniform-address weight quads from LDS, packed FMAs -- with the LDS contents generated in the kernel
// (no global inputs), run many times; every launch must return the bits of the first.  Variants (-D...):
//   (none)      the code as the library had it: hipcc pairs the patch reads into ds_read2_b32 and keeps them in flight with the
//               broadcast ds_read_b128 of the weights behind counted waits
//   VOL         patch reads through a volatile LDS-address-space pointer (single ds_read_b32): the library's fix
//   WLANE       weight quads read at one of TWO addresses (even / odd lanes: not a whole-wave broadcast), patch reads as in (none)
//   NOW         no weight reads at all (constants): only the paired patch reads are in flight
//   -DPROBE_PW=32 / 36 / 64   the patch's row stride (35 in the library: lanes 48-63 then share banks with lanes 0-3; 32: no two
//               lanes of a read share a bank with different addresses; 64: lanes l and l + 32 always do)
//   (FULLWAIT -- an asm s_waitcnt lgkmcnt(0) before every tap -- also stops hipcc from pairing the reads: it tests nothing new)
// Build: hipcc --offload-arch=gfx950 -O3 [-DVOL|-DWLANE] -o probe_lds_mix tools/probes/probe_lds_mix.hip
// Run:   ./probe_lds_mix [launches]      (alone, then next to tools/race_probe.py loaders: tools/probes/run_lds_mix.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifndef PROBE_PW
#define PROBE_PW 35
#endif
constexpr int KC = 16, PW = PROBE_PW, PH = 18, CST = PH * 40;  // (rows of up to 40 floats: PROBE_PW = 32 wraps inside the row)

__global__ __launch_bounds__(256, 2) void probe(float* out, int chunks) {
  __shared__ float xs[KC * CST];
#ifdef WLANE
  __shared__ __attribute__((aligned(16))) float wrep[KC * 9 * 4 * 2];
#else
  __shared__ __attribute__((aligned(16))) float wsm[KC * 9 * 4];
#endif
  const int tid = threadIdx.x;
  const int col = tid & 31, r0 = tid >> 5;
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int ch = 0; ch < chunks; ++ch) {
    __syncthreads();
    for (int e = tid; e < KC * CST; e += 256) xs[e] = (float)((e * 2654435761u + ch * 97u) >> 20) * (1.0f / 4096.0f) - 0.5f;
#ifdef WLANE
    for (int e = tid; e < KC * 9 * 4 * 2; e += 256) {
      const int q = e % (KC * 9 * 4);
      wrep[e] = (float)((q * 40503u + ch * 31u) & 1023) * (1.0f / 1024.0f) - 0.5f;
    }
#else
    for (int e = tid; e < KC * 9 * 4; e += 256) wsm[e] = (float)((e * 40503u + ch * 31u) & 1023) * (1.0f / 1024.0f) - 0.5f;
#endif
    __syncthreads();
#pragma unroll 2
    for (int c = 0; c < KC; ++c) {
      const float* xp = xs + c * CST + r0 * PW + col;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        typedef float f2 __attribute__((ext_vector_type(2)));
#if defined(NOW)
        const float4 wq = make_float4(0.25f + 0.01f * t, -0.5f + 0.02f * c, 0.125f * t, 0.03125f * (c + t));  // no weight reads from LDS
#elif defined(WLANE)
        const float4 wq = *reinterpret_cast<const float4*>(wrep + (tid & 1) * (KC * 9 * 4) + (c * 9 + t) * 4);
#else
        const float4 wq = *reinterpret_cast<const float4*>(wsm + (c * 9 + t) * 4);
#endif
        const f2 w01 = {wq.x, wq.y}, w23 = {wq.z, wq.w};
#ifdef VOL
        const volatile __attribute__((address_space(3))) float* xv = (const volatile __attribute__((address_space(3))) float*)xp;
        const float a0 = xv[(t / 3) * PW + (t % 3)], a1 = xv[(t / 3 + 8) * PW + (t % 3)];
#else
        const float a0 = xp[(t / 3) * PW + (t % 3)], a1 = xp[(t / 3 + 8) * PW + (t % 3)];
#endif
#ifdef FULLWAIT
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        const f2 b0 = {a0, a0}, b1 = {a1, a1};
        f2& c00 = *reinterpret_cast<f2*>(&acc[0][0]);
        f2& c02 = *reinterpret_cast<f2*>(&acc[0][2]);
        f2& c10 = *reinterpret_cast<f2*>(&acc[1][0]);
        f2& c12 = *reinterpret_cast<f2*>(&acc[1][2]);
        c00 = __builtin_elementwise_fma(w01, b0, c00);
        c02 = __builtin_elementwise_fma(w23, b0, c02);
        c10 = __builtin_elementwise_fma(w01, b1, c10);
        c12 = __builtin_elementwise_fma(w23, b1, c12);
      }
    }
  }
  float* o = out + ((size_t)blockIdx.x * 256 + tid) * 8;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) o[4 * i + j] = acc[i][j];
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 300, blocks = 1024, chunks = 4;
  const size_t n = (size_t)blocks * 256 * 8;
  float* d;
  hipMalloc(&d, n * sizeof(float));
  std::vector<float> ref(n), got(n);
  int bad = 0;
  long lanes[4] = {0, 0, 0, 0};
  for (int it = 0; it <= launches; ++it) {
    hipMemset(d, 0, n * sizeof(float));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d, chunks);
    hipMemcpy(it == 0 ? ref.data() : got.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
    if (it == 0) continue;
    if (memcmp(ref.data(), got.data(), n * sizeof(float)) != 0) {
      ++bad;
      for (size_t i = 0; i < n; ++i)
        if (ref[i] != got[i]) lanes[((i / 8) & 63) / 16]++;
    }
  }
  printf("launches differing from the first: %d of %d; differing values by lane quarter (0-15, 16-31, 32-47, 48-63): %ld %ld %ld %ld\n",
         bad, launches, lanes[0], lanes[1], lanes[2], lanes[3]);
  hipFree(d);
  return 0;
}
