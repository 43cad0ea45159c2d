// Access-pattern micro-benchmark (tools/ only, not part of the library): what does the memory system give a conv-like
// tile walk -- 1 workgroup of 4 waves per CU, each tile reading a 16-channel x 18-row x 34-column halo patch per
// K-chunk and writing 64 channels x 16 rows x 32 columns -- with dword accesses (what conv_h2 issues) against
// 16-byte accesses, with no arithmetic in between?   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/_build/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#ifndef MB_H
#define MB_H 256
#define MB_C 64
#endif
constexpr int H = MB_H, W = MB_H, C = MB_C, NB = 16, TH = 16, TW = 32, PH = 18, PW = 34;  // (-DMB_H=32 -DMB_C=512: the deep layers)

// MODE 0: dword loads over the 612 halo positions (tid + 256 k), dword stores, lanes along x
// MODE 1: 16-byte loads of the interior (8 per row) + dword halo columns, 16-byte stores
// LOADS / STORES select which half of the traffic runs
template <int MODE, bool LOADS, bool STORES>
__global__ __launch_bounds__(256, 1) void tile_walk(const float* __restrict__ src, float* __restrict__ dst, int ntiles) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // same XCD-banded order as the conv
    constexpr int NCT = C / 64;
    const int nsp = ntiles / NCT;
    const int grp = tile >> 3, ct = grp % NCT;
    int bid = (tile & 7) * (nsp >> 3) + grp / NCT;
    const int tx = bid % (W / TW); bid /= (W / TW);
    const int ty = bid % (H / TH);
    const int n = bid / (H / TH);
    const float* s = src + (size_t)n * C * H * W;
    float* d = dst + ((size_t)n * C + ct * 64) * H * W;
    float acc = 0.f;
    if (LOADS) {
      for (int q = 0; q < 4; ++q) {  // (four K-chunks per tile whatever C is: the load side is sized for C = 64)
        if (MODE == 0) {
          float v[3][16];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int pos = tid + 256 * k;
            const int py = pos / PW, px = pos - py * PW;
            const int gy = ty * TH - 1 + py, gx = tx * TW - 1 + px;
            const bool ok = pos < PH * PW && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int off = ok ? gy * W + gx : 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) v[k][c] = s[(size_t)(q * 16 + c) * H * W + off];
          }
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc += v[k][c];
        } else if (MODE == 2) {
          // channel-blocked layout [n][c/8][h][w][8]: a halo position's 8 channels are 32 contiguous bytes
          float4 v[3][2][2];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int pos = tid + 256 * k;
            const int py = pos / PW, px = pos - py * PW;
            const int gy = ty * TH - 1 + py, gx = tx * TW - 1 + px;
            const bool ok = pos < PH * PW && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int off = ok ? gy * W + gx : 0;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const float4* bp = reinterpret_cast<const float4*>(s + ((size_t)(q * 2 + g) * H * W + off) * 8);
              v[k][g][0] = bp[0];
              v[k][g][1] = bp[1];
            }
          }
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc += v[k][g][0].x + v[k][g][0].w + v[k][g][1].y + v[k][g][1].z;
        } else {
          // 18 rows x 8 float4 groups = 144 items per channel; 16 channels -> 2304 items / 256 threads = 9 each
          float4 v[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const int item = tid + 256 * k;
            const int c = item / 144, r = item % 144;
            const int py = r / 8, g = r % 8;
            const int gy = min(max(ty * TH - 1 + py, 0), H - 1);
            v[k] = *reinterpret_cast<const float4*>(s + (size_t)(q * 16 + c) * H * W + gy * W + tx * TW + 4 * g);
          }
          // halo columns: 16 ch x 18 rows x 2 = 576 dwords / 256 threads
          float hv[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int item = min(tid + 256 * k, 575);
            const int c = item / 36, r = item % 36;
            const int py = r / 2, side = r % 2;
            const int gy = min(max(ty * TH - 1 + py, 0), H - 1);
            const int gx = min(max(tx * TW + (side ? TW : -1), 0), W - 1);
            hv[k] = s[(size_t)(q * 16 + c) * H * W + gy * W + gx];
          }
#pragma unroll
          for (int k = 0; k < 9; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
#pragma unroll
          for (int k = 0; k < 3; ++k) acc += hv[k];
        }
      }
    }
    if (STORES) {
      const int wave = tid >> 6, lane = tid & 63;
      if (MODE == 0) {
        // wave w: rows 4w..4w+3; lane: x = lane & 31, channel offset 4 * (lane >> 5), as the MFMA layout gives
#pragma unroll 4
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              d[(size_t)co * H * W + (ty * TH + wave * 4 + nt) * W + tx * TW + (lane & 31)] = acc + r;
            }
      } else if (MODE == 2) {
        // blocked layout: lane = pixel x, 4 consecutive channels per register group -> one 16-byte store;
        // a wave instruction writes 32 pixels x 32 bytes = 1 KB contiguous
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            *reinterpret_cast<float4*>(d + ((size_t)cb * H * W + (ty * TH + wave * 4 + nt) * W + tx * TW + (lane & 31)) * 8 +
                                       4 * (lane >> 5)) = make_float4(acc, acc + 1, acc + cb, acc + nt);
      } else {
        // 16-byte stores: lane -> (row-of-8-lanes, 4 consecutive x); a wave instruction writes 8 (channel,row) x 128 B
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const int item = k * 8 + (lane >> 3);            // 0..255: (channel 64) x (row 4)
          const int co = item >> 2, nt = item & 3;
          *reinterpret_cast<float4*>(d + (size_t)co * H * W + (ty * TH + wave * 4 + nt) * W + tx * TW + 4 * (lane & 7)) =
              make_float4(acc, acc + 1, acc + 2, acc + k);
        }
      }
    } else if (acc == 1234.5f) {
      d[tid] = acc;
    }
  }
}

template <int MODE, bool L, bool S>
static void run(const char* name, const float* src, float* dst, int grid, int ntiles, double bytes) {
  const size_t lds = 150 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_walk<MODE, L, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tile_walk<MODE, L, S>), dim3(grid), dim3(256), lds, 0, src, dst, ntiles);
  CK(hipEventRecord(e0));
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((tile_walk<MODE, L, S>), dim3(grid), dim3(256), lds, 0, src, dst, ntiles);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  printf("%-44s grid %5d  %7.1f us  %5.2f TB/s\n", name, grid, ms * 1e3, bytes / ms / 1e9);
}

int main() {
  const size_t n = (size_t)NB * C * H * W;
  float *src, *dst;
  CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&dst, n * 4));
  CK(hipMemset(src, 0, n * 4));
  const int ntiles = NB * (H / TH) * (W / TW) * (C / 64);
  const double b = n * 4.0;
  for (int grid : {ntiles, 256}) {
    run<0, true, true>("dword loads + dword stores", src, dst, grid, ntiles, 2 * b);
    run<1, true, true>("16-byte loads + 16-byte stores", src, dst, grid, ntiles, 2 * b);
    run<2, true, true>("blocked-layout 16-byte loads + stores", src, dst, grid, ntiles, 2 * b);
    run<2, true, false>("blocked-layout loads only", src, dst, grid, ntiles, b);
    run<2, false, true>("blocked-layout stores only", src, dst, grid, ntiles, b);
    run<0, true, false>("dword loads only", src, dst, grid, ntiles, b);
    run<1, true, false>("16-byte loads only", src, dst, grid, ntiles, b);
    run<0, false, true>("dword stores only", src, dst, grid, ntiles, b);
    run<1, false, true>("16-byte stores only", src, dst, grid, ntiles, b);
  }
  return 0;
}
