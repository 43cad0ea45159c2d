// conv_out: the network's last layer (UNet2DModel: conv_norm_out -> SiLU -> conv_out = nn.Conv2d(block_out_channels[0],
// out_channels, 3, padding=1); reference call site DriveSceneGen/scripts/train.py:39-57 -> out_channels 3..8).
//
// The mirror image of conv_in.hip: channel-blocked activations [N][C/8][H][W][8] (fp32 or 16-bit), normalised and activated
// on the way in (GroupNorm scale/shift + SiLU in fp32), become an fp32 [N, Cout <= 8, H, W] image.  97 % of the bytes are the
// read stream, so the kernel is built for it: a workgroup walks 16 x 32 pixel tiles; per 16-channel chunk the 18 x 34 halo
// patch is fetched as whole 32-byte (16-byte) channel blocks one chunk ahead, activated and staged in LDS as
// [position][16 channels] 16-bit, so one ds_read_b128 is one MFMA B operand (K = 1 tap x 16 channels); the weights are the
// A operands (M = cout, zero rows above it) and sit in LDS as ready fragments, built once per workgroup.  A lane ends up
// with one pixel x 4 output channels: 128-byte contiguous stores per wave and channel plane.
// The matrix cores run at 1/8 .. 1/4 row utilisation here and still finish under the read stream's time.
//
// PREC 0 (fp32-equivalent): fp16x2 split (hi*hi + hi*lo + lo*hi, fp32 accumulate); each output channel's weights are scaled
// by the exact power of two of their max |w| first and the result scaled back (no range requirement on w); the activated
// input is O(1) by construction (the call must carry a norm).  PREC 1 / 2: one bf16 / fp16 MFMA per product.
#include "dsg_h16.h"

namespace dsg {

bool prof_on();
int prof_begin(int kid, double flops, double bytes, hipStream_t st);
void prof_end(int idx, hipStream_t st);

struct ConvOutP {
  const void* x;      // [n][cin/8][h][w][8], fp32 or 16-bit
  const float* wt;    // engine layout [cin][9][wstride]
  const float* bias;  // [cout] or NULL
  const float* ss;    // [n][cin][2] GroupNorm scale / shift
  float* dst;         // [n][cout][h][w]
  int n, cin, cout, h, w, wstride, tiles_x, tiles_y, silu;
};

constexpr int CO_TH = 16, CO_TW = 32, CO_PH = CO_TH + 2, CO_PW = CO_TW + 2, CO_NPOS = CO_PH * CO_PW;
constexpr int CO_MAXC = 64;  // channels the weight fragments in LDS are sized for

static bool g_conv_out = true;
void conv_out_set_enabled(int v) { g_conv_out = v != 0; }

__device__ __forceinline__ float co_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int PREC>
__global__ __launch_bounds__(256, PREC == 0 ? 2 : 3) void conv_out_kernel(ConvOutP p) {
  constexpr int NP = PREC == 0 ? 2 : 1;     // operand pieces (hi, lo)
  constexpr int CP = PREC == 0 ? 2 : PREC;  // conversion type of a piece: fp16 for the split
  constexpr int NFRAG = (CO_MAXC / 16) * 9 * 2 * 8;  // (chunk, tap, k-half, cout) weight fragments of 8 channels
  __shared__ __attribute__((aligned(16))) uint4 xs[NP][CO_NPOS][2];  // [piece][position][k-half]: 8 channels each
  __shared__ __attribute__((aligned(16))) uint4 wsm[NP][NFRAG];
  __shared__ unsigned wmx[8];
  __shared__ float wbk[8];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, h2 = l >> 5, r = l & 31;
  const size_t plane = (size_t)p.h * p.w;
  const int ntile = p.tiles_x * p.tiles_y, total = ntile * p.n;
  const int nchunk = p.cin >> 4;

  // ---- weight fragments, once per workgroup: fragment f = ((chunk * 9 + tap) * 2 + half) * 8 + co holds channels
  //      chunk * 16 + half * 8 + 0..7 of (tap, co); PREC 0: scaled per output channel
  if (tid < 8) wmx[tid] = 0u;
  __syncthreads();
  const int nfrag = nchunk * 9 * 2 * 8;
  float wv8[3][8];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int f = tid + 256 * k;
    const int co = f & 7, half = (f >> 3) & 1, ct = f >> 4, tap = ct % 9, chunk = ct / 9;
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = chunk * 16 + half * 8 + j;
      const bool ok = f < nfrag && co < p.cout;
      const float v = p.wt[((size_t)(ok ? ci : 0) * 9 + tap) * p.wstride + (ok ? co : 0)];
      wv8[k][j] = ok ? v : 0.f;
      m = fmaxf(m, fabsf(wv8[k][j]));
    }
    if (PREC == 0 && f < nfrag) atomicMax(&wmx[co], __float_as_uint(m));  // (non-negative floats order like their bits)
  }
  __syncthreads();
  if (tid < 8) {
    float back = 1.f;
    if constexpr (PREC == 0) {
      const unsigned b = wmx[tid];
      int e = (int)(b >> 23) - 127;
      if (b == 0u || b >= 0x7F800000u) e = 0;
      e = e < -100 ? -100 : (e > 100 ? 100 : e);
      back = __uint_as_float((unsigned)(127 + e) << 23);
    }
    wbk[tid] = back;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int f = tid + 256 * k;
    if (f < nfrag) {
      const float sw = PREC == 0 ? 1.0f / wbk[f & 7] : 1.f;  // (an exact power of two)
      unsigned hi[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = wv8[k][2 * j] * sw, b = wv8[k][2 * j + 1] * sw;
        hi[j] = pack2<CP>(a, b);
        if constexpr (PREC == 0) lo[j] = pack2<CP>(a - lo16<CP>(hi[j]), b - hi16<CP>(hi[j]));
      }
      wsm[0][f] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      if constexpr (PREC == 0) wsm[1][f] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
  }
  // (visible after the first barrier of the chunk loop)
  const int cl0 = 4 * h2;  // this lane's output channels cl0 .. cl0 + 3 (accumulator registers 0..3)
  float bs[4], back[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bs[i] = (p.bias && cl0 + i < p.cout) ? p.bias[cl0 + i] : 0.f;
    back[i] = 1.f;
  }

  // ---- raw channel blocks of one (tile, chunk): up to 3 positions per thread, two 8-channel blocks each; every load
  //      unconditional (clamped address), fetched one chunk ahead of its use
  constexpr int RW = PREC == 0 ? 2 : 1;  // 16-byte words per 8-channel block
  uint4 raw[3][2][RW];
  auto fetch = [&](int t, int chunk) {
    const int n = t / ntile, tin = t - n * ntile;
    const int ty = tin / p.tiles_x, tx = tin - ty * p.tiles_x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int pos = min(tid + 256 * k, CO_NPOS - 1);
      const int py = pos / CO_PW, px = pos - py * CO_PW;
      const int gy = min(max(ty * CO_TH - 1 + py, 0), p.h - 1), gx = min(max(tx * CO_TW - 1 + px, 0), p.w - 1);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const size_t blk = ((size_t)n * (p.cin >> 3) + 2 * chunk + b) * plane + (size_t)gy * p.w + gx;
        const uint4* sp = reinterpret_cast<const uint4*>(p.x) + blk * RW;
#pragma unroll
        for (int q = 0; q < RW; ++q) raw[k][b][q] = sp[q];
      }
    }
  };

  if ((int)blockIdx.x < total) fetch(blockIdx.x, 0);
#pragma unroll 1
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int n = t / ntile, tin = t - n * ntile;
    const int ty = tin / p.tiles_x, tx = tin - ty * p.tiles_x;
    const int oy0 = ty * CO_TH, ox0 = tx * CO_TW;
    f32x16 acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[s][v] = 0.f;

#pragma unroll 1
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      // ---- activate and stage the chunk's patch
      __syncthreads();  // (the previous chunk's readers are done)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int pos = tid + 256 * k;
        if (pos < CO_NPOS) {
          const int py = pos / CO_PW, px = pos - py * CO_PW;
          const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
          const bool ok = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            float v[8];
            if constexpr (PREC == 0) {
              const uint4 q0 = raw[k][b][0], q1 = raw[k][b][RW - 1];
              v[0] = __uint_as_float(q0.x); v[1] = __uint_as_float(q0.y); v[2] = __uint_as_float(q0.z); v[3] = __uint_as_float(q0.w);
              v[4] = __uint_as_float(q1.x); v[5] = __uint_as_float(q1.y); v[6] = __uint_as_float(q1.z); v[7] = __uint_as_float(q1.w);
            } else {
              const uint4 q0 = raw[k][b][0];
              v[0] = lo16<CP>(q0.x); v[1] = hi16<CP>(q0.x); v[2] = lo16<CP>(q0.y); v[3] = hi16<CP>(q0.y);
              v[4] = lo16<CP>(q0.z); v[5] = hi16<CP>(q0.z); v[6] = lo16<CP>(q0.w); v[7] = hi16<CP>(q0.w);
            }
            const float* ssb = p.ss + ((size_t)n * p.cin + chunk * 16 + b * 8) * 2;  // (uniform: scalar loads)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float a = v[j] * ssb[2 * j] + ssb[2 * j + 1];
              if (p.silu) a = co_silu(a);
              v[j] = ok ? a : 0.f;
            }
            unsigned hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              hi[j] = pack2<CP>(v[2 * j], v[2 * j + 1]);
              if constexpr (PREC == 0) lo[j] = pack2<CP>(v[2 * j] - lo16<CP>(hi[j]), v[2 * j + 1] - hi16<CP>(hi[j]));
            }
            xs[0][pos][b] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            if constexpr (PREC == 0) xs[1][pos][b] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
          }
        }
      }
      __syncthreads();
      // ---- the next chunk's (or the next tile's first chunk's) raw blocks, issued ahead of this chunk's arithmetic
      {
        const bool last = chunk + 1 == nchunk;
        fetch(last ? min(t + (int)gridDim.x, total - 1) : t, last ? 0 : chunk + 1);
      }
      // ---- this chunk's weights as A operands (rows >= 8 of the 32-row tile are zero)
      half8 wa[NP][9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int f = ((chunk * 9 + tap) * 2 + h2) * 8 + (r & 7);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          uint4 w = wsm[q][f];
          if (r >= 8) w = make_uint4(0u, 0u, 0u, 0u);
          wa[q][tap] = __builtin_bit_cast(half8, w);
        }
      }
      // ---- 4 rows of 32 pixels per wave, 9 taps each
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int y = 4 * wv + s;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int dy = tap / 3, dx = tap - 3 * dy;
          const int pos = (y + dy) * CO_PW + r + dx;
          const half8 bh = __builtin_bit_cast(half8, xs[0][pos][h2]);
          if constexpr (PREC == 0) {
            const half8 bl = __builtin_bit_cast(half8, xs[1][pos][h2]);
            acc[s] = mma16<2>(wa[1][tap], bh, acc[s]);
            acc[s] = mma16<2>(wa[0][tap], bl, acc[s]);
            acc[s] = mma16<2>(wa[0][tap], bh, acc[s]);
          } else {
            acc[s] = mma16<PREC>(wa[0][tap], bh, acc[s]);
          }
        }
      }
    }

    // ---- epilogue: accumulator registers 0..3 of lane (h2, r) are output channels 4 h2 + 0..3 of pixel r
    if constexpr (PREC == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) back[i] = wbk[(cl0 + i) & 7];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const size_t pix = (size_t)(oy0 + 4 * wv + s) * p.w + ox0 + r;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (cl0 + i < p.cout) p.dst[((size_t)n * p.cout + cl0 + i) * plane + pix] = acc[s][i] * back[i] + bs[i];
    }
  }
}

// shapes the kernel takes: 3x3, stride 1, one channel-blocked source with a norm in the call, cin a multiple of 16 up to 64,
// at most 8 output channels written as an fp32 [N,C,H,W] image, 16 x 32 pixel tiles
bool conv_out_eligible(const dsg_conv_args* a, int hout, int wout) {
  if (!g_conv_out) return false;
  return a->ksize == 3 && a->stride == 1 && !a->upsample && !a->pool2 && a->c1 == 0 && a->c0 % 16 == 0 && a->c0 <= CO_MAXC &&
         a->src_layout == 1 && a->dst_layout == 0 && a->gn_scale_shift && !a->temb && !a->residual && !a->stats_out &&
         a->cout <= 8 && hout % CO_TH == 0 && wout % CO_TW == 0;
}

int conv_out_launch(const dsg_conv_args* a, int hout, int wout, hipStream_t st) {
  ConvOutP p;
  p.x = a->src0; p.wt = a->weight; p.bias = a->bias; p.ss = a->gn_scale_shift; p.dst = a->dst;
  p.n = a->n; p.cin = a->c0; p.cout = a->cout; p.h = hout; p.w = wout; p.silu = a->silu;
  p.wstride = a->weight_cout_stride ? a->weight_cout_stride : a->cout;
  p.tiles_x = wout / CO_TW; p.tiles_y = hout / CO_TH;
  int pi = -1;
  if (prof_on()) {
    const double px = (double)a->n * hout * wout;
    pi = prof_begin(12, 2.0 * px * a->cout * a->c0 * 9,
                    (a->compute_dtype ? 2.0 : 4.0) * px * a->c0 + 4.0 * (9.0 * a->c0 * a->cout + px * a->cout), st);
  }
  const int total = p.tiles_x * p.tiles_y * p.n;
  // persistent over tiles: as many workgroups as are resident at once (2 per CU with the split's registers and LDS, 3 otherwise)
  const int cap = a->compute_dtype == DSG_F32 ? 512 : 768;
  const int gx = total < cap ? total : cap;
  if (a->compute_dtype == DSG_F32) hipLaunchKernelGGL(conv_out_kernel<0>, dim3(gx), dim3(256), 0, st, p);
  else if (a->compute_dtype == DSG_BF16) hipLaunchKernelGGL(conv_out_kernel<1>, dim3(gx), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(conv_out_kernel<2>, dim3(gx), dim3(256), 0, st, p);
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

}  // namespace dsg
