// conv_in: the network's first layer (UNet2DModel.conv_in, diffusers unet_2d.py: nn.Conv2d(in_channels, block_out_channels[0],
// 3, padding=1); reference call site DriveSceneGen/scripts/train.py:39-57 -> in_channels 3..8).
//
// An fp32 [N, Cin <= 8, H, W] image becomes channel-blocked activations [N][Cout/8][H][W][8] (fp32, or the 16-bit type of the
// mixed-precision modes).  The work is 0.3 % of a forward's FLOPs and 2 % of its bytes, so the kernel is built for the
// HBM write stream: a workgroup takes 16 x 32 pixel tiles x 32 or 64 output channels; the 18 x 34 halo patch is staged
// once in LDS as [position][8 channels] 16-bit (one ds_read_b128 = one MFMA B operand: K = 2 taps x 8 channels), the
// weights live in registers as MFMA A operands (M = cout), so a lane ends up with 4 consecutive channels of one pixel:
// 16-byte (fp32) / 8-byte (16-bit) stores, a wave writes 32 pixels x 32 bytes contiguous per channel block.
// The epilogue also leaves the per-tile GroupNorm statistics of what it wrote ([n][cout][tiles][2] fp64, fixed order), which
// the first resnet's norm1 used to get from a pass of its own over the tensor.
//
// PREC 0 (fp32-equivalent): every product is the fp16x2 split (hi*hi + hi*lo + lo*hi, fp32 accumulate) of operands that
// were first scaled by exact powers of two -- the patch by its own max |x|, each output channel's weights by their
// max |w| -- and the result scaled back: no range requirement on x or w (the guard of DESIGN 4.4 is built in).
// PREC 1 / 2: x and w rounded once to bf16 / fp16, one MFMA per product, as torch.autocast does for this conv.
#include "dsg_h16.h"

namespace dsg {

bool prof_on();
int prof_begin(int kid, double flops, double bytes, hipStream_t st);
void prof_end(int idx, hipStream_t st);

struct ConvInP {
  const float* x;     // [n][cin][h][w]
  const float* wt;    // engine layout [cin][9][wstride]
  const float* bias;  // [cout] or NULL
  void* dst;          // [n][cout/8][h][w][8], fp32 or 16-bit
  double* stats;      // optional [n][cout][tiles][2]
  int n, cin, cout, h, w, wstride, tiles_x, tiles_y;
};

constexpr int CI_TH = 16, CI_TW = 32, CI_PH = CI_TH + 2, CI_PW = CI_TW + 2, CI_NPOS = CI_PH * CI_PW;

static bool g_conv_in = true;
static int g_conv_in_epoch = 0;
void conv_in_set_enabled(int v) { g_conv_in = v != 0; ++g_conv_in_epoch; }
int conv_in_tuning_epoch() { return g_conv_in_epoch; }

__device__ __forceinline__ float pow2_scale_of(float m, float* back) {
  // s = 2^-e, *back = 2^e with e = exponent of m clamped to [-100, 100]; m == 0 / inf / nan -> 1
  const unsigned b = __float_as_uint(m) & 0x7FFFFFFFu;
  int e = (int)(b >> 23) - 127;
  if (b == 0u || b >= 0x7F800000u) e = 0;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  *back = __uint_as_float((unsigned)(127 + e) << 23);
  return __uint_as_float((unsigned)(127 - e) << 23);
}

// WS: wave groups along cout.  WS 2: waves 0-1 own the workgroup's first 32 output channels, waves 2-3 the next 32, 8 rows
// each (one staged patch serves 64 channels); WS 1: 32 channels per workgroup, 4 rows per wave.
// Workgroups are persistent over tiles (grid.x < tiles): the weight operands are built once per workgroup.
// x + (x of the lane a DPP control selects); five of them make a fixed-order sum over each half of the wave on the VALU
// (the cross-lane shuffles of __shfl_xor are LDS-pipe instructions: 320 per wave and tile bounded the first version)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float ci_dpp_add(float x) {
  const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, true);
  return x + __int_as_float(y);
}
// afterwards lanes 16..31 hold the sum over lanes 0..31, lanes 48..63 the sum over lanes 32..63
__device__ __forceinline__ float ci_half_wave_sum(float x) {
  x = ci_dpp_add<0xB1>(x);        // quad_perm [1,0,3,2]
  x = ci_dpp_add<0x4E>(x);        // quad_perm [2,3,0,1]
  x = ci_dpp_add<0x141>(x);       // row_half_mirror
  x = ci_dpp_add<0x140>(x);       // row_mirror
  x = ci_dpp_add<0x142, 0xA>(x);  // row_bcast15 into rows 1 and 3
  return x;
}

template <int PREC, int WS>
__global__ __launch_bounds__(256, PREC == 0 ? 2 : 3) void conv_in_kernel(ConvInP p) {
  constexpr int NP = PREC == 0 ? 2 : 1;     // operand pieces (hi, lo)
  constexpr int CP = PREC == 0 ? 2 : PREC;  // conversion type of a piece: fp16 for the split
  constexpr int ROWS = 4 * WS, WPG = 4 / WS;  // rows per wave, waves per cout group
  __shared__ __attribute__((aligned(16))) uint4 xs[NP][CI_NPOS];
  __shared__ float red[4];
  __shared__ float sred[4][32][2];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, h2 = l >> 5, r = l & 31;
  const int grp = wv / WPG, wrow0 = (wv % WPG) * ROWS;
  const int cbase = (blockIdx.y * WS + grp) * 32;
  const size_t plane = (size_t)p.h * p.w;
  const int ntile = p.tiles_x * p.tiles_y, total = ntile * p.n;

  // ---- the weights of this wave's 32 output channels as A operands: slot (ks, h2, j) = (tap 2 ks + h2, channel j);
  //      every load is unconditional (clamped address), the padding slots are zeroed afterwards
  half8 wa[NP][5];
  float wback = 1.f;  // (PREC 0) 2^e of output channel cbase + r
  {
    const int co = cbase + r;
    float wv8[5][8];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int tap = 2 * ks + h2, tapc = tap < 9 ? tap : 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) wv8[ks][j] = p.wt[((size_t)(j < p.cin ? j : 0) * 9 + tapc) * p.wstride + co];
    }
    float m = 0.f;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (2 * ks + h2 >= 9 || j >= p.cin) wv8[ks][j] = 0.f;
        m = fmaxf(m, fabsf(wv8[ks][j]));
      }
    float sw = 1.f;
    if constexpr (PREC == 0) {
      m = fmaxf(m, __shfl_xor(m, 32));
      sw = pow2_scale_of(m, &wback);
    }
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      unsigned hi[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = wv8[ks][2 * j] * sw, b = wv8[ks][2 * j + 1] * sw;
        hi[j] = pack2<CP>(a, b);
        if constexpr (PREC == 0) lo[j] = pack2<CP>(a - lo16<CP>(hi[j]), b - hi16<CP>(hi[j]));
      }
      wa[0][ks] = __builtin_bit_cast(half8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
      if constexpr (PREC == 0) wa[1][ks] = __builtin_bit_cast(half8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
    }
  }
  float bs[4][4], wbk[4][4];  // bias and weight scale-back of this lane's channels 8 j + 4 h2 + i
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cl = 8 * j + 4 * h2 + i;
      bs[j][i] = p.bias ? p.bias[cbase + cl] : 0.f;
      wbk[j][i] = PREC == 0 ? __shfl(wback, cl) : 1.f;  // (lane cl holds channel cl's factor)
    }
  const int cblocks = p.cout >> 3;

  // the raw patch of a tile: up to 3 positions per thread, 8 channel slots each, every load unconditional (clamped
  // addresses; what lies outside the image or beyond cin is zeroed in registers).  The NEXT tile's patch is fetched while
  // the current one is being computed: issued after this tile's stores it would wait in the memory pipeline behind them.
  float xv[3][8];
  auto fetch = [&](int t) {
    const int n = t / ntile, tin = t - n * ntile;
    const int ty = tin / p.tiles_x, tx = tin - ty * p.tiles_x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int pos = min(tid + 256 * k, CI_NPOS - 1);
      const int py = pos / CI_PW, px = pos - py * CI_PW;
      const int gy = ty * CI_TH - 1 + py, gx = tx * CI_TW - 1 + px;
      const bool ok = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
      const float* sp = p.x + (size_t)n * p.cin * plane + (size_t)min(max(gy, 0), p.h - 1) * p.w + min(max(gx, 0), p.w - 1);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float v = sp[(size_t)(c < p.cin ? c : 0) * plane];
        xv[k][c] = (ok && c < p.cin) ? v : 0.f;
      }
    }
  };
  if ((int)blockIdx.x < total) fetch(blockIdx.x);
#pragma unroll 1
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int n = t / ntile, tin = t - n * ntile;
    const int ty = tin / p.tiles_x, tx = tin - ty * p.tiles_x;
    const int oy0 = ty * CI_TH, ox0 = tx * CI_TW;
    // ---- stage it (PREC 0: scaled by the power of two of its own max)
    float sx = 1.f, xback = 1.f;
    if constexpr (PREC == 0) {
      float m = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 8; ++c) m = fmaxf(m, fabsf(xv[k][c]));
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      __syncthreads();  // (the previous tile's readers of red / xs are done)
      if (l == 0) red[wv] = m;
      __syncthreads();
      m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      sx = pow2_scale_of(m, &xback);
    } else {
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (tid + 256 * k < CI_NPOS) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = xv[k][2 * j] * sx, b = xv[k][2 * j + 1] * sx;
          hi[j] = pack2<CP>(a, b);
          if constexpr (PREC == 0) lo[j] = pack2<CP>(a - lo16<CP>(hi[j]), b - hi16<CP>(hi[j]));
        }
        xs[0][tid + 256 * k] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        if constexpr (PREC == 0) xs[1][tid + 256 * k] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
    }
    __syncthreads();
    fetch(min(t + (int)gridDim.x, total - 1));  // (the last round re-reads a tile it does not use)

    // ---- ROWS rows of 32 pixels per wave
    float ssum[16], ssq[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) ssum[v] = ssq[v] = 0.f;
#pragma unroll 1
    for (int s = 0; s < ROWS; ++s) {
      const int y = wrow0 + s;
      f32x16 acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const int tap = (2 * ks + h2) < 9 ? (2 * ks + h2) : 8;  // (slot 9 has zero weights: it re-reads tap 8's values)
        const int dy = tap / 3, dx = tap - 3 * dy;
        const int pos = (y + dy) * CI_PW + r + dx;
        const half8 bh = __builtin_bit_cast(half8, xs[0][pos]);
        if constexpr (PREC == 0) {
          const half8 bl = __builtin_bit_cast(half8, xs[1][pos]);
          acc = mma16<2>(wa[1][ks], bh, acc);
          acc = mma16<2>(wa[0][ks], bl, acc);
          acc = mma16<2>(wa[0][ks], bh, acc);
        } else {
          acc = mma16<PREC>(wa[0][ks], bh, acc);
        }
      }
      const size_t pix = (size_t)(oy0 + y) * p.w + ox0 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cb = (cbase >> 3) + j;
        const size_t off = (((size_t)n * cblocks + cb) * plane + pix) * 8 + 4 * h2;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[4 * j + i] * (wbk[j][i] * xback) + bs[j][i];
        if constexpr (PREC == 0) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.dst) + off) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          const unsigned w0 = pack2<PREC>(v[0], v[1]), w1 = pack2<PREC>(v[2], v[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.dst) + off) = make_uint2(w0, w1);
          v[0] = lo16<PREC>(w0); v[1] = hi16<PREC>(w0); v[2] = lo16<PREC>(w1); v[3] = hi16<PREC>(w1);  // (as stored)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ssum[4 * j + i] += v[i];
          ssq[4 * j + i] += v[i] * v[i];
        }
      }
    }

    // ---- per-tile statistics: over the 32 pixel lanes, then over the group's waves in wave order, in fp64 at the end
    if (p.stats != nullptr) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const float a = ci_half_wave_sum(ssum[v]), q = ci_half_wave_sum(ssq[v]);
        if (r == 31) {
          const int cl = 8 * (v >> 2) + 4 * h2 + (v & 3);
          sred[wv][cl][0] = a;
          sred[wv][cl][1] = q;
        }
      }
      __syncthreads();
      if (tid < 64 * WS) {
        const int g = tid >> 6, cl = (tid & 63) >> 1, which = tid & 1;
        double tsum = 0.0;
#pragma unroll
        for (int k = 0; k < WPG; ++k) tsum += (double)sred[g * WPG + k][cl][which];
        p.stats[(((size_t)n * p.cout + (blockIdx.y * WS + g) * 32 + cl) * ntile + tin) * 2 + which] = tsum;
      }
    }
  }
}

// shapes the kernel takes: 3x3, stride 1, one fp32 [N,C,H,W] source of at most 8 channels with no norm / temb / residual /
// pool in the call, a channel-blocked result, 16 x 32 pixel tiles, cout in 32-channel tiles inside the weight stride
bool conv_in_eligible(const dsg_conv_args* a, int hout, int wout) {
  if (!g_conv_in) return false;
  const int cin = a->c0 + a->c1;
  const int wstride = a->weight_cout_stride ? a->weight_cout_stride : a->cout;
  return a->ksize == 3 && a->stride == 1 && !a->upsample && !a->pool2 && a->c1 == 0 && cin <= 8 && a->src_layout == 0 &&
         a->dst_layout == 1 && !a->gn_scale_shift && !a->temb && !a->residual && a->cout % 32 == 0 &&
         wstride >= a->cout && hout % CI_TH == 0 && wout % CI_TW == 0;
}

int conv_in_stats_tiles(const dsg_conv_args* a, int hout, int wout) {
  return conv_in_eligible(a, hout, wout) ? (hout / CI_TH) * (wout / CI_TW) : 0;
}

template <int PREC, int WS>
static void conv_in_go(const ConvInP& p, hipStream_t st) {
  // persistent over tiles: as many workgroups as are resident at once (2 per CU with the split's registers, 3 otherwise),
  // an equal share of tiles each where a nearby grid size divides them
  const int total = p.tiles_x * p.tiles_y * p.n, gy = p.cout / (32 * WS);
  int gx = total;
  const int cap = (PREC == 0 ? 512 : 768) / (gy < 4 ? gy : 4);
  if (gx > cap) {
    gx = cap;
    while (gx > cap - cap / 8 && total % gx) --gx;
    if (total % gx) gx = cap;
  }
  hipLaunchKernelGGL((conv_in_kernel<PREC, WS>), dim3(gx, gy), dim3(256), 0, st, p);
}

int conv_in_launch(const dsg_conv_args* a, int hout, int wout, hipStream_t st) {
  ConvInP p;
  p.x = a->src0; p.wt = a->weight; p.bias = a->bias; p.dst = a->dst; p.stats = a->stats_out;
  p.n = a->n; p.cin = a->c0; p.cout = a->cout; p.h = hout; p.w = wout;
  p.wstride = a->weight_cout_stride ? a->weight_cout_stride : a->cout;
  p.tiles_x = wout / CI_TW; p.tiles_y = hout / CI_TH;
  int pi = -1;
  if (prof_on()) {
    const double px = (double)a->n * hout * wout;
    pi = prof_begin(11, 2.0 * px * a->cout * a->c0 * 9,
                    4.0 * (px * a->c0 + 9.0 * a->c0 * a->cout) + (a->compute_dtype ? 2.0 : 4.0) * px * a->cout, st);
  }
  const bool ws2 = a->cout % 64 == 0;
  if (a->compute_dtype == DSG_F32) ws2 ? conv_in_go<0, 2>(p, st) : conv_in_go<0, 1>(p, st);
  else if (a->compute_dtype == DSG_BF16) ws2 ? conv_in_go<1, 2>(p, st) : conv_in_go<1, 1>(p, st);
  else ws2 ? conv_in_go<2, 2>(p, st) : conv_in_go<2, 1>(p, st);
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

}  // namespace dsg
