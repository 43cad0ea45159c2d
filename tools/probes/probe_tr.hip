// Probe: what does ds_read_b64_tr_b16 deliver?  LDS holds u16 value = its own element index; every lane passes an
// address; the two result dwords per lane are dumped.  Patterns: (a) lane-linear 8-byte addresses; (b) rows of 16 u16
// (32-byte row stride): lane l -> row (l & 15), 8-byte column (l >> 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned* out, int pattern) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int elem = 0;
  if (pattern == 0) elem = l * 4;
  else if (pattern == 1) elem = (l & 15) * 16 + (l >> 4) * 4;
  else if (pattern == 2) elem = (l & 15) * 64 + (l >> 4) * 4;      // 128-byte rows
  else elem = (l & 3) * 4 + (l >> 2) * 16;                            // 4 lanes cover a 16-element row
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)(lds + elem);
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  u2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  out[(pattern * 64 + l) * 2 + 0] = v.x;
  out[(pattern * 64 + l) * 2 + 1] = v.y;
}

int main() {
  unsigned* d;
  hipMalloc(&d, 4 * 64 * 2 * 4);
  for (int p = 0; p < 4; ++p) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
  std::vector<unsigned> h(4 * 64 * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  for (int p = 0; p < 4; ++p) {
    printf("pattern %d\n", p);
    for (int l = 0; l < 64; ++l)
      printf("  lane %2d: %4u %4u %4u %4u\n", l, h[(p * 64 + l) * 2] & 0xffff, h[(p * 64 + l) * 2] >> 16,
             h[(p * 64 + l) * 2 + 1] & 0xffff, h[(p * 64 + l) * 2 + 1] >> 16);
  }
  return 0;
}
