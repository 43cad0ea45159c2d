// Probe (round 4): ds_read2_b32 under LDS contention from other processes, with hand-placed waits.
// Every thread streams pairs of neighbouring floats out of a 40-KB LDS table (stride 35 floats between the half-waves, as in
// conv_fewout_kernel) through a two-deep software pipeline and sums them; the sum is known exactly.  Modes:
//   0  ds_read2_b32 (offset1 = offset0 + 1), counted waits: s_waitcnt lgkmcnt(1) before consuming the older pair
//   1  ds_read2_b32, full waits: s_waitcnt lgkmcnt(0) before every use
//   2  two ds_read_b32 per pair, counted waits (lgkmcnt(2))
//   3  ds_read2_b32 with offset1 = offset0 + 35 (two rows, not neighbours), counted waits
//   4  MIXED: a far pair (offset1 = offset0 + 218) issued first, a neighbouring pair second, counted wait, the FAR pair consumed first
//      (what the compiler's loop has in flight: pairs of one 8-byte piece next to pairs of two pieces)
//   5  the same with the neighbouring pair first
//   7  a pair by ds_read2_b32, FULL wait, then v_pk_fma_f32 with op_sel:[0,1,0] (both lanes take the pair's HIGH dword)
//   8  the same with op_sel_hi:[1,0,1] (both lanes take the LOW dword)
//   9  ds_read2_b32, full wait, the HIGH dword consumed by the very next instruction (v_add_f32)
//  10  the same with s_nop 7 between the wait and the consumer
//  11  ds_read2_b64, full wait, the LAST dword consumed by the very next instruction
//  12  ds_read_b64 (one 8-byte piece), full wait, the high dword consumed at once
//  13  NO LDS at all: the pair made by VALU moves, then v_pk_fma_f32 op_sel:[0,1,0]
//  14  ... v_pk_mul_f32 op_sel:[0,1,0] + v_pk_add_f32      15  ... v_pk_add_f32 op_sel:[0,1,0]
//  16  NO LDS: v_pk_fma_f16 op_sel:[0,1,0] (the 16-bit packed form: both result halves from src1's HIGH half)
//  17  NO LDS: v_pk_mul_f16 op_sel_hi:[1,0] (both from the LOW half: control)
//  18  ONE process, ONE kernel: waves 1 and 3 of every workgroup loop v_mfma_f32_32x32x16_f16 (AGPR accumulators) while waves 0 and 2
//      run mode 13's v_pk_fma_f32 op_sel:[0,1,0] -- is another PROCESS needed, or only another WAVE on the SIMD?
//   6  SIX pairs in flight (alternating neighbouring / far), consumed oldest first behind lgkmcnt(5), (4), .. (0)
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_lds_read2 tools/probes/probe_lds_read2.hip
// Run:   ./probe_lds_read2 [launches]   (alone, then next to loader processes: tools/probes/run_lds_mix.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int N = 10080;  // floats in the table
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
  __shared__ float xs[N];
  const int tid = threadIdx.x;
  for (int e = tid; e < N; e += 256) xs[e] = (float)((e * 2654435761u) >> 22);  // small integers: sums are exact in fp32
  __syncthreads();
  const int col = tid & 31, r0 = tid >> 5;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(xs + r0 * 35 + col);
  float acc = 0.f;
  f2 p0, p1;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  f32x16 m0 = {}, m1 = {};
  half8 fa, fb;
  for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(0.001f * tid); fb[j] = (_Float16)(0.002f * j); }
  const bool mfma_wave = MODE == 18 && ((tid >> 6) & 1);
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 18) {
      if (mfma_wave) {   // (wave-uniform branch)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\t"
                     "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1"
                     : "+a"(m0), "+a"(m1) : "v"(fa), "v"(fb));
        continue;
      }
    }
    unsigned a = base + (unsigned)((it * 37) % 9000) * 4;   // (uniform step: the same address pattern every iteration)
    unsigned b = a + 630 * 4;
    if constexpr (MODE == 16 || MODE == 17) {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      h2 w = {(_Float16)1.0f, (_Float16)2.0f}, pr, r, z = {(_Float16)0.f, (_Float16)0.f};
      const float lo = (float)((it * 7 + tid) & 255), hi = (float)((it * 13 + tid * 3) & 511);
      pr.x = (_Float16)lo;
      pr.y = (_Float16)hi;
      asm volatile("" : "+v"(pr));
      if constexpr (MODE == 16) asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(w), "v"(pr), "v"(z));
      else asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(w), "v"(pr));
      asm volatile("s_nop 4" ::: "memory");
      acc += (float)r.x + 0.5f * (float)r.y - (MODE == 16 ? hi : lo);   // = + the selected half, exactly
    } else if constexpr (MODE == 13 || MODE == 14 || MODE == 15 || MODE == 18) {
      f2 w = {1.0f, 2.0f}, acc2 = {acc, 0.f}, pr;
      pr.x = (float)((it * 7 + tid) & 511);
      pr.y = (float)((it * 13 + tid * 3) & 1023);
      asm volatile("" : "+v"(pr));
      if constexpr (MODE == 13 || MODE == 18) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc2) : "v"(w), "v"(pr));
      else if constexpr (MODE == 14) {
        f2 t2;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t2) : "v"(w), "v"(pr));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc2) : "v"(t2));
      } else {
        f2 t2;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t2) : "v"(w), "v"(pr));   // (1 + y, 2 + y)
        acc2.x += t2.x - 1.0f;
        acc2.y += 2.0f * (t2.y - 2.0f);
      }
      asm volatile("s_nop 4" ::: "memory");
      acc = acc2.x + 0.5f * acc2.y - pr.y;
    } else if constexpr (MODE == 9 || MODE == 10) {
      asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:37" : "=v"(p0) : "v"(a) : "memory");
      if constexpr (MODE == 9) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tv_add_f32 %0, %0, %1" : "+v"(acc) : "v"(p0.y) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\tv_add_f32 %0, %0, %1" : "+v"(acc) : "v"(p0.y) : "memory");
    } else if constexpr (MODE == 11) {
      typedef float f4v __attribute__((ext_vector_type(4)));
      f4v q;
      const unsigned a8 = a & ~7u;   // (8-byte aligned)
      asm volatile("ds_read2_b64 %0, %1 offset0:1 offset1:18" : "=v"(q) : "v"(a8) : "memory");
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tv_add_f32 %0, %0, %1" : "+v"(acc) : "v"(q.w) : "memory");
    } else if constexpr (MODE == 12) {
      const unsigned a8 = a & ~7u;
      asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(p0) : "v"(a8) : "memory");
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tv_add_f32 %0, %0, %1" : "+v"(acc) : "v"(p0.y) : "memory");
    } else if constexpr (MODE == 7 || MODE == 8) {
      f2 w = {1.0f, 2.0f}, acc2 = {acc, 0.f};
      asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:37" : "=v"(p0) : "v"(a) : "memory");
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if constexpr (MODE == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc2) : "v"(w), "v"(p0));
      else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2) : "v"(w), "v"(p0));
      asm volatile("s_nop 4" ::: "memory");
      acc = acc2.x + 0.5f * acc2.y - (MODE == 7 ? p0.y : p0.x);   // = acc + 1 * v + 0.5 * (2 * v) - v = acc + v, exactly (small integers)
    } else if constexpr (MODE == 6) {
      f2 q0, q1, q2, q3, q4, q5;
      asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(q0) : "v"(a) : "memory");
      asm volatile("ds_read2_b32 %0, %1 offset0:8 offset1:226" : "=v"(q1) : "v"(a) : "memory");
      asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(q2) : "v"(a) : "memory");
      asm volatile("ds_read2_b32 %0, %1 offset0:35 offset1:220" : "=v"(q3) : "v"(a) : "memory");
      asm volatile("ds_read2_b32 %0, %1 offset0:36 offset1:37" : "=v"(q4) : "v"(a) : "memory");
      asm volatile("ds_read2_b32 %0, %1 offset0:70 offset1:250" : "=v"(q5) : "v"(a) : "memory");
      asm volatile("s_waitcnt lgkmcnt(5)\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(q0.x), "v"(q0.y));
      asm volatile("s_waitcnt lgkmcnt(4)\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(q1.x), "v"(q1.y));
      asm volatile("s_waitcnt lgkmcnt(3)\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(q2.x), "v"(q2.y));
      asm volatile("s_waitcnt lgkmcnt(2)\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(q3.x), "v"(q3.y));
      asm volatile("s_waitcnt lgkmcnt(1)\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(q4.x), "v"(q4.y));
      asm volatile("s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(q5.x), "v"(q5.y));
    } else if constexpr (MODE == 4 || MODE == 5) {
      if constexpr (MODE == 4) {
        asm volatile("ds_read2_b32 %0, %1 offset0:8 offset1:226" : "=v"(p0) : "v"(a) : "memory");
        asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(p1) : "v"(b) : "memory");
      } else {
        asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(p0) : "v"(b) : "memory");
        asm volatile("ds_read2_b32 %0, %1 offset0:8 offset1:226" : "=v"(p1) : "v"(a) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
      asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(p0.x), "v"(p0.y));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(p1.x), "v"(p1.y));
    } else if constexpr (MODE == 0 || MODE == 1 || MODE == 3) {
      if constexpr (MODE == 3) {
        asm volatile("ds_read2_b32 %0, %1 offset1:35" : "=v"(p0) : "v"(a) : "memory");
        asm volatile("ds_read2_b32 %0, %1 offset1:35" : "=v"(p1) : "v"(b) : "memory");
      } else {
        asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(p0) : "v"(a) : "memory");
        asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(p1) : "v"(b) : "memory");
      }
      if constexpr (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
      asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(p0.x), "v"(p0.y));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(p1.x), "v"(p1.y));
    } else {
      asm volatile("ds_read_b32 %0, %1" : "=v"(s0) : "v"(a) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(s1) : "v"(a) : "memory");
      asm volatile("ds_read_b32 %0, %1" : "=v"(s2) : "v"(b) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(s3) : "v"(b) : "memory");
      asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(s0), "v"(s1));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc) : "v"(s2), "v"(s3));
    }
  }
  if (mfma_wave) {
    acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += (m0[r] + m1[r]) * 0.f;   // (keeps the matrix results alive; contributes exactly 0 or NaN-free junk * 0)
    acc = -1.0f;                                                   // marker: not a checked value
  }
  out[(size_t)blockIdx.x * 256 + tid] = acc;
}

template <int MODE>
static void run(int launches, const char* what) {
  const int blocks = 1024, iters = 600;
  const size_t n = (size_t)blocks * 256;
  float* d;
  hipMalloc(&d, n * sizeof(float));
  std::vector<float> ref(n), got(n);
  {  // the exact sums (small integers)
    std::vector<float> xs(N);
    for (int e = 0; e < N; ++e) xs[e] = (float)(((unsigned)e * 2654435761u) >> 22);
    const int second = MODE == 3 ? 35 : 1;
    for (int tid = 0; tid < 256; ++tid) {
      const int b0 = (tid >> 5) * 35 + (tid & 31);
      double sum = 0;
      for (int it = 0; it < iters; ++it) {
        const int a = b0 + (it * 37) % 9000, b = a + 630;
        if (MODE == 18) sum += (float)((it * 13 + tid * 3) & 1023);
        else if (MODE == 16) sum += (float)((it * 13 + tid * 3) & 511);
        else if (MODE == 17) sum += (float)((it * 7 + tid) & 255);
        else if (MODE >= 13 && MODE <= 15) sum += (float)((it * 13 + tid * 3) & 1023);
        else if (MODE == 9 || MODE == 10) sum += xs[a + 37];
        else if (MODE == 11) sum += xs[(a & ~1) + 2 * 18 + 1];
        else if (MODE == 12) sum += xs[(a & ~1) + 2 + 1];
        else if (MODE == 7) sum += xs[a + 37];
        else if (MODE == 8) sum += xs[a + 2];
        else if (MODE == 6) sum += xs[a] + xs[a + 1] + xs[a + 8] + xs[a + 226] + xs[a + 2] + xs[a + 3] + xs[a + 35] + xs[a + 220] + xs[a + 36] + xs[a + 37] + xs[a + 70] + xs[a + 250];
        else if (MODE == 4) sum += xs[a + 8] + xs[a + 226] + xs[b] + xs[b + 1];
        else if (MODE == 5) sum += xs[b] + xs[b + 1] + xs[a + 8] + xs[a + 226];
        else sum += xs[a] + xs[a + second] + xs[b] + xs[b + second];
      }
      if (MODE == 18 && ((tid >> 6) & 1)) sum = -1.0;   // (the matrix waves' marker)
      for (int blk = 0; blk < blocks; ++blk) ref[(size_t)blk * 256 + tid] = (float)sum;
    }
  }
  int bad = 0;
  long lanes[4] = {0, 0, 0, 0};
  for (int it = 1; it <= launches; ++it) {
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipMemcpy(got.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
    bool diff = false;
    for (size_t i = 0; i < n; ++i)
      if (ref[i] != got[i]) {
        if (!diff && bad == 0 && getenv("PROBE_VERBOSE")) printf("  first mismatch: index %zu (tid %zu) got %.1f want %.1f\n", i, i & 255, got[i], ref[i]);
        diff = true;
        lanes[(i & 63) / 16]++;
      }
    bad += diff;
  }
  printf("mode %d (%s): launches with wrong sums %d of %d; by lane quarter %ld %ld %ld %ld\n", MODE, what, bad, launches,
         lanes[0], lanes[1], lanes[2], lanes[3]);
  hipFree(d);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 200;
  run<0>(launches, "ds_read2_b32 neighbours, counted waits");
  run<1>(launches, "ds_read2_b32 neighbours, full waits");
  run<2>(launches, "2 x ds_read_b32, counted waits");
  run<3>(launches, "ds_read2_b32 rows apart, counted waits");
  run<4>(launches, "far pair then neighbouring pair in flight, counted wait, far consumed first");
  run<5>(launches, "neighbouring pair then far pair in flight, counted wait, neighbouring consumed first");
  run<7>(launches, "ds_read2_b32, full wait, v_pk_fma_f32 op_sel:[0,1,0]");
  run<8>(launches, "ds_read2_b32, full wait, v_pk_fma_f32 op_sel_hi:[1,0,1]");
  run<9>(launches, "ds_read2_b32, full wait, high dword consumed by the next instruction");
  run<10>(launches, "ds_read2_b32, full wait, s_nop 7, high dword consumed");
  run<11>(launches, "ds_read2_b64, full wait, last dword consumed by the next instruction");
  run<12>(launches, "ds_read_b64, full wait, high dword consumed by the next instruction");
  run<13>(launches, "no LDS: v_pk_fma_f32 op_sel:[0,1,0] on a VALU-made pair");
  run<14>(launches, "no LDS: v_pk_mul_f32 op_sel:[0,1] + v_pk_add_f32");
  run<15>(launches, "no LDS: v_pk_add_f32 op_sel:[0,1]");
  run<16>(launches, "no LDS: v_pk_fma_f16 op_sel:[0,1,0]");
  run<17>(launches, "no LDS: v_pk_mul_f16 op_sel_hi:[1,0]");
  run<18>(launches, "ONE kernel: waves 1, 3 loop the 16-deep MFMA, waves 0, 2 v_pk_fma_f32 op_sel:[0,1,0]");
  run<6>(launches, "six pairs in flight, consumed oldest first behind lgkmcnt(5) .. (0)");
  return 0;
}
