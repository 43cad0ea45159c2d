// 3x3 stride-1 convolution with fp32-equivalent accuracy on the fp16 matrix cores ("fp16x2 split").
//
// Same contract, tiling and fusions as conv.hip's conv_mfma_kernel (reference call sites: every 3x3
// Conv2d of diffusers' UNet2DModel as built at DriveSceneGen/scripts/train.py:39-57 and run at
// DriveSceneGen/pipeline/training_pipeline.py:84), but the contraction runs at the 16x-faster f16 MFMA
// rate without giving up fp32 accuracy:
//
//   x = x1 + x2 * 2^-11,  x1 = fp16(x),  x2 = fp16((x - x1) * 2^11)      (|x - x1 - x2*2^-11| <= 2^-24 |x|)
//   w = w1 + w2 * 2^-11   likewise (split once, at weight re-layout time)
//   sum w*x  ~=  sum w1*x1  +  2^-11 * sum (w1*x2 + w2*x1)              (dropped w2*x2 term: 2^-24 relative)
//
// i.e. 3 v_mfma_f32_32x32x16_f16 per 16-deep k-step instead of 8 v_mfma_f32_32x32x2_f32: 5.3x fewer matrix
// cycles.  fp16 x fp16 products are exact in the fp32 accumulator; the scaled low-order products go to a
// second accumulator so that nothing is lost to fp16's narrow exponent (the 2^11 pre-scale keeps the low
// parts normal).  Measured error vs fp64 is at or below that of a sequential fp32 fmaf chain
// (tests/test_gpu_ops.py::test_conv_h2_*).  Inputs must satisfy |x| < 65504 (GroupNorm/SiLU outputs and
// residual-stream activations do).
//
// LDS images (per K-chunk of 16 channels, double-buffered; same bytes as the fp32 kernel's):
//   X[piece 2][g 2][pos 10x34][8 halfs]   -- lane = pixel reads one 16-B fragment (k-group g = lane>>5)
//   W[piece 2][tap 9][g 2][cout 64][8]    -- lane = cout  reads one 16-B fragment; filled by LDS-DMA
//                                            (global_load_lds_dwordx4: the pre-split weights need no math)
// A and B use the same (g, j) <-> channel 8g+j map, so the MFMA's internal k order is irrelevant.
#include "dsg_common.h"
#include <algorithm>
#include <type_traits>

namespace dsg {

bool prof_on();
int prof_begin(int kid, double flops, double bytes, hipStream_t st);
void prof_end(int idx, hipStream_t st);

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct ConvH2P {
  const float* src0;
  const float* src1;
  int c0, c1, cin;
  int n, hin, win;
  int hc, wc;
  int hout, wout;
  int cout, cout_pad;
  int wh_stride;       // couts per weight row (>= cout_pad when `wh` is a column window of a wider matrix)
  const _Float16* wh;  // [cin/16][2][9][2][wh_stride][8]
  const float* bias;
  const float* ss;
  int silu;
  const float* temb;
  int temb_stride;
  const float* res;
  float* dst;
  double* stats;  // optional [n][cout][hout/8 * wout/32][2]: per-tile (sum, sum of squares) of the values written
  int tiles_x, tiles_y;
};

constexpr int H2_TW = 32, H2_KC = 16, H2_BM = 64;

// NT = output rows per wave (2 or 4): a workgroup covers 4*NT rows x 32 cols.  NT = 4 halves the LDS operand
// traffic per MFMA (each weight fragment feeds 4 pixel tiles) and the weight DMA per MFMA; it needs 256
// accumulator registers (the kernel owns the SIMD: 1 wave, 512 registers).
// KS = 3 (halo of 1) or 1 (no halo; attention projections and resnet shortcuts)
// NW = waves per workgroup (4: one per SIMD with the whole register file; 8: two per SIMD with half of it each,
// so that one wave's staging / LDS / wait time is covered by the other's MFMAs)
template <int NT, int KS, int NW = 4, int TAPS_ = KS * KS, int BM_ = 64>
struct H2Geom {
  static constexpr int BM = BM_;                      // output channels per workgroup: 64, or 32 (two workgroups per CU)
  static constexpr int NTH = 64 * NW;
  static constexpr int TAPS = TAPS_;                  // 4 in the folded up-sampler mode (2x2 taps of the 3x3 patch)
  static constexpr int TH = NW * NT;
  static constexpr int PH = TH + KS - 1;
  static constexpr int PW = H2_TW + KS - 1;
  static constexpr int PSZ = PW * PH;                 // KS=3: 340 (NT=2) / 612 (NT=4); KS=1: 256 / 512
  static constexpr int WHALFS = 2 * TAPS * 2 * BM * 8;  // [piece][tap][g][cout][8]: 36864 B / 4096 B at BM = 64
  static constexpr int XHALFS = 2 * 2 * PSZ * 8;      // [piece][g][pos][8]
  static constexpr int BUF_BYTES = (WHALFS + XHALFS) * 2 + 64;  // + a dump slot for masked lanes
  static constexpr int FULL = PSZ / NTH;              // full NTH-position slabs per k-group
  static constexpr bool HAS_REM = (PSZ % NTH) != 0;   // KS=3 leaves a remainder slab shared by the two k-groups
  static constexpr int NU = 2 * FULL + (HAS_REM ? 1 : 0);  // staging units per thread
  static constexpr int REM0 = FULL * NTH;             // first position of the remainder unit
  static constexpr int NSEG = 4 * TAPS;               // (piece, tap, g) weight segments per chunk, BM x 16 bytes each
  static constexpr int NUNIT = NSEG * BM / 64;        // 1-KB DMA units per chunk (a unit = 64 / BM segments)
  static constexpr int NDMA = (NUNIT + NW - 1) / NW;  // weight DMAs per wave per chunk
  static_assert(PSZ - REM0 <= NTH / 2, "the remainder slab must fit half the workgroup per k-group");
};

// x + (x of the lane selected by a DPP control): the building block of a fixed-order 32-lane tree sum
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float x) {
  const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, true);
  return x + __int_as_float(y);
}
// after this, lanes 16..31 hold the sum over lanes 0..31 and lanes 48..63 the sum over lanes 32..63
__device__ __forceinline__ float half_wave_sum(float x) {
  x = dpp_add<0xB1>(x);        // quad_perm [1,0,3,2]
  x = dpp_add<0x4E>(x);        // quad_perm [2,3,0,1]
  x = dpp_add<0x141>(x);       // row_half_mirror
  x = dpp_add<0x140>(x);       // row_mirror: every lane of a 16-row holds the row sum
  x = dpp_add<0x142, 0xA>(x);  // row_bcast15 into rows 1 and 3
  return x;
}

__device__ __forceinline__ float silu_fast_h(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// GM: 3 STRIDE-2 3x3 conv (Downsample2D) as a 2x2 conv over the space-to-depth image, which for channel-blocked
// sources is pure addressing: k-group (cb, py, px) of the 4C "channels" is channel block cb read at pixels
// (2y + py, 2x + px); output row oy needs input rows 2oy-1 (y' = oy-1, py = 1), 2oy (oy, 0), 2oy+1 (oy, 1), i.e. the
// 2x2 corner {y'-1, y'} x {x'-1, x'} of the low-resolution patch with zero weights where no 3x3 tap lands (7 of the
// 16 (tap, phase) pairs): 16/9 of the useful products, on the pipe that is 5x faster than the f32 one.
// GM: 0 plain, 1 nearest x2 gather, 2 nearest x2 FOLDED into the weights: Upsample2D + 3x3 conv is four 2x2 convs of
// the low-resolution input, one per output phase (py, px) = (Y & 1, X & 1): rows {y-1: W0, y: W1+W2} for py = 0 and
// {y: W0+W1, y+1: W2} for py = 1, likewise in x -- 16 tap products per input pixel instead of 36.  The kernel runs on
// the low-resolution grid with the phase as an extra (outer) cout-tile index, walks the phase's 2x2 corner of the 3x3
// patch and scatters its results to the (2y+py, 2x+px) pixels.
// NT: rows per wave; KS: 3 | 1.
// Staging units are arranged so that the k-group g (hence the channel plane and the GroupNorm scale/shift) of
// every unit is WAVE-UNIFORM: channel-plane bases and scale/shift live in SGPRs (s_load / saddr-form global
// loads), and the only per-lane address is the 32-bit halo offset computed once per tile.
//   units 0..FULL-1: g = 0, halo positions tid + 256*i      units FULL..2*FULL-1: g = 1, same positions
//   last unit: g = wave >> 1, halo position FULL*256 + (tid & 127)   (the remainder, valid where < PSZ)
// ACT: 0 the input is used as it is; 2 GroupNorm affine + SiLU; 3 decided at run time from p.ss / p.silu
// OCC: workgroups the kernel is compiled to fit per CU (register budget 512 / (OCC * NW / 4) per lane)
// LAY: bit 0: the sources are channel-blocked [N][C/8][H][W][8] (a halo position's k-group is 32 contiguous bytes:
//      two 16-byte loads instead of eight dword gathers from eight channel planes); bit 1: dst / residual are
//      (a lane's four consecutive output channels are one 16-byte store; a wave instruction writes 1 KB contiguous)
// BM: output channels per workgroup.  32 (with NT = 2, OCC = 2) is the small-workgroup geometry for the shallow levels:
//     80 KB of LDS and half the register file, so two workgroups share a CU and one's patch loads and output stores
//     run under the other's MFMAs (a workgroup that owns the CU runs those phases back to back).
template <int GM, int NT, int KS, int ACT = 3, int NW = 4, int OCC = 1, int LAY = 0, int BM = 64>
__global__ __launch_bounds__(64 * NW, OCC * NW / 4) void conv_h2_kernel(ConvH2P p) {
  constexpr bool SB = (LAY & 1) != 0, DB = (LAY & 2) != 0;
  constexpr int MTN = BM / 32;  // 32-channel MFMA tiles per workgroup
  using G = H2Geom<NT, KS, NW, (GM == 2 || GM == 3) ? 4 : KS * KS, BM>;
  constexpr int NTH = G::NTH;
  constexpr int H2_TH = G::TH, H2_PSZ = G::PSZ, H2_XHALFS = G::XHALFS, H2_BUF_BYTES = G::BUF_BYTES, H2_NU = G::NU;
  constexpr int FULL = G::FULL, TAPS = G::TAPS, H2_PW = G::PW, H2_WHALFS = G::WHALFS, PADK = KS / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
#ifdef DSG_H2_TIMING
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;

  // Workgroup id -> (spatial tile, cout tile).  Consecutive ids go to the 8 XCDs in turn, each with its own L2.
  // XCD k gets a CONTIGUOUS eighth of the spatial tiles, walked in raster order with the cout tiles of one patch
  // on neighbouring ids: workgroups that share input -- the same patch for another cout tile, or the 128-byte
  // lines and halo rows a patch has in common with its left/right/upper/lower neighbours -- run at the same time
  // behind the same L2, so that data comes from HBM once instead of once per XCD.
  const int nct = (p.cout_pad / BM) * (GM == 2 ? 4 : 1), nsp = p.tiles_x * p.tiles_y * p.n;
  int bid, ct;
  if ((nsp & 7) == 0) {
    const int grp = blockIdx.x >> 3;
    ct = grp % nct;
    bid = (blockIdx.x & 7) * (nsp >> 3) + grp / nct;
  } else {
    ct = blockIdx.x % nct;
    bid = blockIdx.x / nct;
  }
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int phase = GM == 2 ? ct / (p.cout_pad / BM) : 0;  // (py, px) = (phase >> 1, phase & 1)
  if (GM == 2) ct -= phase * (p.cout_pad / BM);
  const int m0 = ct * BM;
  const int oy0 = ty * H2_TH, ox0 = tx * H2_TW;
  const int plane = p.hin * p.win;
  const int nq = p.cin / H2_KC;
  const int g2 = wave / (NW / 2);  // k-group of the remainder unit (uniform per wave)

  // Per staging unit: global halo offset, and the LDS slots of its two pieces.  Positions outside the image (zero
  // padding) or past the patch write to a dump slot instead; the real slots of padding positions are zeroed once.
  int goff[H2_NU], xoff[H2_NU], xoff2[H2_NU], zoff[H2_NU];
#pragma unroll
  for (int i = 0; i < H2_NU; ++i) {
    const int g = i < FULL ? 0 : (i < 2 * FULL ? 1 : g2);
    const int pos = i < 2 * FULL ? tid + NTH * (i % FULL) : G::REM0 + (tid & (NTH / 2 - 1));
    int off = 0, xo = H2_WHALFS + H2_XHALFS, xo2 = H2_WHALFS + H2_XHALFS, zo = -1;
    if (pos < H2_PSZ) {
      const int py = pos / H2_PW, px = pos - py * H2_PW;
      const int gy = oy0 - PADK + py, gx = ox0 - PADK + px;
      const int slot = H2_WHALFS + (g * H2_PSZ + pos) * 8;  // piece 0; piece 1 is 2*PSZ*8 halfs further
      if (gy >= 0 && gy < p.hc && gx >= 0 && gx < p.wc) {
        off = GM == 3 ? (2 * gy) * p.win + 2 * gx : (GM == 1 ? (gy >> 1) : gy) * p.win + (GM == 1 ? (gx >> 1) : gx);
        xo = slot;
        xo2 = slot + 2 * H2_PSZ * 8;
      } else {
        zo = slot;
      }
    }
    goff[i] = off;
    xoff[i] = xo;
    xoff2[i] = xo2;
    zoff[i] = zo;
  }
  const bool has_ss = ACT == 3 ? p.ss != nullptr : ACT != 0;
  const bool do_silu = ACT == 3 ? (has_ss && p.silu) : ACT == 2;
  const float* ssg = has_ss ? p.ss + (size_t)n * p.cin * 2 : nullptr;

  float xr[H2_NU][8];
  // GroupNorm (scale, shift) of this image's channels: copied once into LDS behind the two K-chunk buffers; a commit
  // reads its 8 channels from there (uniform address: a broadcast read) instead of carrying them in registers
  float* ssl = reinterpret_cast<float*>(smem_raw + 2 * H2_BUF_BYTES);

  auto src_of = [&](int q) -> const float* {  // uniform
    const int cb = q * H2_KC;
    return (cb < p.c0) ? p.src0 + ((size_t)n * p.c0 + cb) * plane
                       : p.src1 + ((size_t)n * p.c1 + (cb - p.c0)) * plane;
  };
  auto unit_g = [&](int i) -> int { return i < FULL ? 0 : (i < 2 * FULL ? 1 : g2); };
  // patch loads are buffer loads: descriptor = the chunk's 16 channel planes (uniform), soffset = the channel
  // plane (uniform), voffset = the lane's halo offset (fixed for the whole tile) -- no per-load address math
  int soff[8];  // byte offsets of the 8 channel planes of a k-group: loop-invariant SGPRs
#pragma unroll
  for (int j = 0; j < 8; ++j) soff[j] = __builtin_amdgcn_readfirstlane(j * plane * 4);
  // descriptor of k-group g of chunk q: its 8 channel planes / its channel block (uniform)
  auto grp_rs = [&](const float* sp, int q, int g) -> __amdgpu_buffer_rsrc_t {
    if constexpr (GM == 3) {  // group (cb, py, px) of the space-to-depth image: block cb, first pixel (py, px)
      const int gi = 2 * q + g, cb = gi >> 2, pp = gi & 3;
      const int first = ((pp >> 1) * p.win + (pp & 1)) * 8;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src0 + ((size_t)n * p.c0 + cb * 8) * plane + first), 0,
                                               (8 * plane - first) * 4, 0x00020000);
    } else {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sp + (size_t)(g * 8) * plane), 0, 8 * plane * 4,
                                               0x00020000);
    }
  };
  auto load_unit_to = [&](float (&dst)[H2_NU][8], int i, const float* sp, int q) {
    const __amdgpu_buffer_rsrc_t rs = grp_rs(sp, q, unit_g(i));
    if constexpr (SB) {  // (the k-group's 8 planes and its channel block start at the same address)
      const float4 lo = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i] * 32, 0, 0));
      const float4 hi = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i] * 32, 16, 0));
      dst[i][0] = lo.x; dst[i][1] = lo.y; dst[i][2] = lo.z; dst[i][3] = lo.w;
      dst[i][4] = hi.x; dst[i][5] = hi.y; dst[i][6] = hi.z; dst[i][7] = hi.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        dst[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, goff[i] * 4, soff[j], 0));
    }
  };
  auto load_unit = [&](int i, int q, const float* sp) { load_unit_to(xr, i, sp, q); };
  auto commit_unit_from = [&](const float (&src)[H2_NU][8], int i, int q, unsigned char* buf) {  // q: chunk staged
    half8 h1, h2;
    float4 sr[4];
    if (has_ss) {
      const float4* ssq = reinterpret_cast<const float4*>(ssl + 2 * (q * H2_KC + unit_g(i) * 8));
#pragma unroll
      for (int j = 0; j < 4; ++j) sr[j] = ssq[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = src[i][j];
      if (has_ss) v = (j & 1) ? v * sr[j / 2].y + sr[j / 2].w : v * sr[j / 2].x + sr[j / 2].z;
      const float sv = silu_fast_h(v);
      v = do_silu ? sv : v;
      const _Float16 a = (_Float16)v;
      h1[j] = a;
      h2[j] = (_Float16)((v - (float)a) * 2048.0f);
    }
    // (branch-free: a branch here would fence the instruction scheduler between staging and MFMAs)
    _Float16* xb = reinterpret_cast<_Float16*>(buf);
    *reinterpret_cast<half8*>(xb + xoff[i]) = h1;
    *reinterpret_cast<half8*>(xb + xoff2[i]) = h2;
  };
  // The same work in quarter-unit steps, so that a K-chunk's staging can be dealt out evenly over its taps: step P
  // turns two channels (2jp, 2jp+1) of unit P/4 into fp16 pairs -- the unit's LDS write rides on its last step --
  // and refills the two registers with chunk q+2's values.
  half8 h1s[H2_NU], h2s[H2_NU];
  auto stage_step = [&](int P, int qs, unsigned char* buf, bool stage, bool load, const float* spn) {  // (loads: chunk qs + 1)
    const int i = P / 4, jp = P % 4;
    if (stage) {
      float4 s4 = make_float4(1.f, 1.f, 0.f, 0.f);
      if (has_ss) s4 = *reinterpret_cast<const float4*>(ssl + 2 * (qs * H2_KC + unit_g(i) * 8 + 2 * jp));
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * jp + e;
        float v = xr[i][j];
        if (has_ss) v = e ? v * s4.y + s4.w : v * s4.x + s4.z;
        const float sv = silu_fast_h(v);
        v = do_silu ? sv : v;
        const _Float16 a = (_Float16)v;
        h1s[i][j] = a;
        h2s[i][j] = (_Float16)((v - (float)a) * 2048.0f);
      }
      if (jp == 3) {
        _Float16* xb = reinterpret_cast<_Float16*>(buf);
        *reinterpret_cast<half8*>(xb + xoff[i]) = h1s[i];
        *reinterpret_cast<half8*>(xb + xoff2[i]) = h2s[i];
      }
    }
    if (load) {
      const __amdgpu_buffer_rsrc_t rs = grp_rs(spn, qs + 1, unit_g(i));
      if constexpr (SB) {  // four channels are free after every second step: one 16-byte load refills them
        if (jp & 1) {
          const float4 v4 =
              __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i] * 32, 8 * (jp - 1), 0));
          xr[i][2 * jp - 2] = v4.x; xr[i][2 * jp - 1] = v4.y; xr[i][2 * jp] = v4.z; xr[i][2 * jp + 1] = v4.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e)
          xr[i][2 * jp + e] =
              __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, goff[i] * 4, soff[2 * jp + e], 0));
      }
    }
  };
  auto commit_unit = [&](int i, int q, unsigned char* buf) { commit_unit_from(xr, i, q, buf); };
  // weight slab of chunk q: 36 segments (piece, tap, g) of 64 couts x 16 B, moved global -> LDS by DMA;
  // wave w moves segments w, w+4, ...
  // Addressing is split so that a DMA costs one 64-bit scalar add: the tile's weight base and the byte offset of each
  // of this wave's segments are loop-invariant scalars, the chunk offset is added once per chunk by the caller, and
  // the only vector operand is the constant lane * 16.
  const unsigned segb = (unsigned)p.wh_stride * 16u;  // bytes of one (piece, tap, g) segment row in global memory
  const unsigned chunkb = G::NSEG * segb;             // bytes of one K-chunk's weights
  const char* wtile = reinterpret_cast<const char*>(p.wh + ((size_t)phase * nq * G::NSEG * p.wh_stride + m0) * 8);
  // a DMA moves 1 KB = 64 / BM segments of BM couts x 16 B: LDS [segment][cout][8 halfs] is contiguous, in global
  // memory the segments are `segb` apart
  int segoff[G::NDMA];
#pragma unroll
  for (int k = 0; k < G::NDMA; ++k)
    segoff[k] = __builtin_amdgcn_readfirstlane(min(wave + NW * k, G::NUNIT - 1) * (64 / BM) * (int)segb);
  const int lane16 = (lane % BM) * 16 + (lane / BM) * (int)segb;
  auto dma_weights = [&](int k, const char* wq, unsigned char* buf) {  // wq: wtile + chunk * chunkb (uniform)
    // (uniform; a wave whose last share falls past the end repeats the final unit: same bytes, no branch)
    const int unit = min(wave + NW * k, G::NUNIT - 1);
    // Issued as inline asm on purpose: hipcc's wait-count pass cannot tell the DMA's LDS destination (the other
    // buffer) from the fragment reads of this one, and with a DMA it knows of in flight it puts vmcnt(0) -- a wait
    // for every outstanding patch load as well -- in front of each following ds_read.  Untracked VMEM operations
    // only make the compiler's own counted vmcnt(N) waits stricter (the counter retires in order); the DMA's
    // completion is waited for explicitly before the chunk's closing barrier.
    const unsigned lds_addr =
        (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(buf + unit * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n" ::"v"(lane16), "s"(wq + segoff[k]),
                 "s"(__builtin_amdgcn_readfirstlane(lds_addr))  // (uniform by construction)
                 : "memory");
  };

  f32x16 acc_hi[MTN][NT], acc_lo[MTN][NT];
#pragma unroll
  for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc_hi[mt][nt][r] = 0.f;
        acc_lo[mt][nt][r] = 0.f;
      }

  unsigned char* buf0 = smem_raw;
  unsigned char* buf1 = smem_raw + H2_BUF_BYTES;

  // zero padding: halo positions outside the image are zeroed once in both buffers and never written again
#pragma unroll
  for (int i = 0; i < H2_NU; ++i) {
    if (zoff[i] >= 0) {
      half8 z;
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
      _Float16* b0 = reinterpret_cast<_Float16*>(buf0);
      _Float16* b1 = reinterpret_cast<_Float16*>(buf1);
      *reinterpret_cast<half8*>(b0 + zoff[i]) = z;
      *reinterpret_cast<half8*>(b0 + zoff[i] + 2 * H2_PSZ * 8) = z;
      *reinterpret_cast<half8*>(b1 + zoff[i]) = z;
      *reinterpret_cast<half8*>(b1 + zoff[i] + 2 * H2_PSZ * 8) = z;
    }
  }
  // prologue: chunk 0 -> buffer 0; chunk 1 -> registers.  Everything that goes to memory is issued first and
  // together (both chunks' patches, the weight DMAs, the scale/shift table), so the tile pays one memory round
  // trip before its first MFMA, not one per dependent step.
#ifdef DSG_H2_TIMING
  unsigned long long rt_p[4];
#define DSG_PT(i) do { __builtin_amdgcn_sched_barrier(0); rt_p[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
  DSG_PT(0);
#else
#define DSG_PT(i)
#endif
  {
    // Issue order: weight DMAs, chunk 0's patch (into a scratch set), chunk 1's patch (into xr, where the K loop
    // expects it).  The tile waits only for the DMAs and chunk 0 -- vmcnt retires in order, so a counted wait with
    // chunk 1's loads still outstanding covers exactly those -- and chunk 1 lands under chunk 0's MFMAs.
    float xr0[H2_NU][8];
    float ssv[2048 / NTH];  // this thread's share of the image's scale/shift table (cin <= 1024): oldest loads
    if (has_ss) {
#pragma unroll
      for (int k = 0; k < 2048 / NTH; ++k) ssv[k] = ssg[min(tid + NTH * k, 2 * p.cin - 1)];
    }
#pragma unroll
    for (int k = 0; k < G::NDMA; ++k) dma_weights(k, wtile, buf0);
    const float* sp = src_of(0);
#pragma unroll
    for (int i = 0; i < H2_NU; ++i) load_unit_to(xr0, i, sp, 0);
    if (nq > 1) {
      const float* sp1 = src_of(1);
#pragma unroll
      for (int i = 0; i < H2_NU; ++i) load_unit_to(xr, i, sp1, 1);
    }
    DSG_PT(1);
    if (has_ss) {
#pragma unroll
      for (int k = 0; k < 2048 / NTH; ++k)
        if (tid + NTH * k < 2 * p.cin) {
          // global [c][scale | shift] -> LDS per channel PAIR (sc0, sc1, sh0, sh1): a staging step's two channels
          // then take their scales and shifts as register pairs (one packed FMA, no shuffling moves)
          const int idx = tid + NTH * k, c = idx >> 1, which = idx & 1;
          ssl[4 * (c >> 1) + 2 * which + (c & 1)] = ssv[k];
        }
      __syncthreads();  // the scale/shift table is in LDS
    }
    DSG_PT(2);
#pragma unroll
    for (int i = 0; i < H2_NU; ++i) commit_unit_from(xr0, i, 0, buf0);
    DSG_PT(3);
  }
  // (the DMAs were issued before every patch load: once chunk 0's values have been used they have landed; 8 * NU
  // loads of chunk 1 may still be in flight)
  if (nq > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SB ? 2 : 8) * H2_NU) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // One K-chunk: MFMAs on `cur`; STAGE: chunk q+1 (patch in registers, weights by DMA) goes into `nxt`;
  // LOAD: chunk q+2's patch is fetched into the registers just freed.
#ifdef DSG_H2_TIMING
  unsigned long long t_tap[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_vm = 0, t_bar = 0;
  const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();  // 100 MHz, same base on every CU
#endif
  auto chunk = [&](int q, auto stage_tag, auto load_tag) {
    constexpr bool STAGE = decltype(stage_tag)::value, LOAD = decltype(load_tag)::value;
    unsigned char* cur = (q & 1) ? buf1 : buf0;
    unsigned char* nxt = (q & 1) ? buf0 : buf1;
    const float* spn = LOAD ? src_of(q + 2) : nullptr;
    const char* wqn = wtile + (size_t)(q + 1) * chunkb;  // the staged chunk's weights
    const _Float16* wl = reinterpret_cast<const _Float16*>(cur);
    const _Float16* xl = wl + H2_WHALFS;
    // Operand fragments are fetched one tap ahead into the other half of fa/fb: the reads of tap t+1 are issued
    // BEFORE tap t's staging writes in program order, so tap t's MFMAs depend on registers only and the scheduler
    // is free to interleave them with the staging work (LDS reads after a possibly-aliasing LDS write are not).
    half8 fa[2][MTN][2], fb[2][NT][2];  // [parity][tile][piece]
    auto load_frags = [&](int tap, int par) {
      // folded up-sampler: the phase's 2x2 corner of the patch; stride 2: the {y-1, y} x {x-1, x} corner
      const int dy = GM == 2 ? (phase >> 1) + (tap >> 1) : (GM == 3 ? (tap >> 1) : tap / KS);
      const int dx = GM == 2 ? (phase & 1) + (tap & 1) : (GM == 3 ? (tap & 1) : tap % KS);
#pragma unroll
      for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
          fa[par][mt][pc] =
              *reinterpret_cast<const half8*>(wl + (((pc * TAPS + tap) * 2 + half) * BM + mt * 32 + l31) * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
          fb[par][nt][pc] = *reinterpret_cast<const half8*>(
              xl + ((pc * 2 + half) * H2_PSZ + (wave * NT + nt + dy) * H2_PW + l31 + dx) * 8);
    };
    load_frags(0, 0);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      __builtin_amdgcn_sched_barrier(0);
#ifdef DSG_H2_TIMING
      const unsigned long long tt0 = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
#endif
      if (tap + 1 < TAPS) load_frags(tap + 1, (tap + 1) & 1);
      if (KS == 1) {  // one tap: all units and the four weight segments ride on it
#pragma unroll
        for (int u = 0; u < H2_NU; ++u) {
          if (STAGE) commit_unit(u, q + 1, nxt);
          if (LOAD) load_unit(u, q + 2, spn);
        }
        if (STAGE) dma_weights(0, wqn, nxt);  // (KS = 1: NSEG = 4 <= NW)
      }
      // KS = 3: the chunk's staging steps and weight DMAs are dealt out evenly over taps 0..TAPS-2 (the last tap
      // stays clear so that the newest loads have a tap's worth of MFMAs to land before the closing vmcnt(0))
      if (KS == 3 && tap < TAPS - 1) {
        constexpr int NSTEP = 4 * H2_NU, ST = TAPS - 1;
#pragma unroll
        for (int P = tap * NSTEP / ST; P < (tap + 1) * NSTEP / ST; ++P) stage_step(P, q + 1, nxt, STAGE, LOAD, spn);
        if (STAGE) {
#pragma unroll
          for (int k = tap * G::NDMA / ST; k < (tap + 1) * G::NDMA / ST; ++k) dma_weights(k, wqn, nxt);
        }
      }
      const int par = tap & 1;
#pragma unroll
      for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc_hi[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][mt][0], fb[par][nt][0], acc_hi[mt][nt], 0, 0, 0);
          acc_lo[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][mt][0], fb[par][nt][1], acc_lo[mt][nt], 0, 0, 0);
          acc_lo[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][mt][1], fb[par][nt][0], acc_lo[mt][nt], 0, 0, 0);
        }
      // Issue order within the tap: with one wave per SIMD nothing else fills the matrix pipe while this wave
      // issues staging work, so spread that work between the MFMAs (at most ~5 issues hide behind one MFMA)
      // instead of leaving it in one block as the scheduler would.
      if (KS == 3 && tap < TAPS - 1 && (STAGE || LOAD)) {
#pragma unroll
        for (int m = 0; m < 3 * MTN * NT; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // 2 VALU
        }
      }
#ifdef DSG_H2_TIMING
      __builtin_amdgcn_sched_barrier(0);
      t_tap[tap] += __builtin_readcyclecounter() - tt0;
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef DSG_H2_TIMING  // tools/ only: where does a wave wait at the end of a chunk?  (p.stats = 4 counters)
    const unsigned long long ta = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long tb = __builtin_readcyclecounter();
    __builtin_amdgcn_s_barrier();
    const unsigned long long tc = __builtin_readcyclecounter();
    t_vm += tb - ta;
    t_bar += tc - tb;
    __builtin_amdgcn_sched_barrier(0);
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the weight DMAs (not tracked by the compiler) have landed
    __syncthreads();  // nxt is complete; everyone is done reading cur
#endif
  };
  using T = std::true_type;
  using F = std::false_type;
#ifdef DSG_H2_TIMING
  const unsigned long long t_begin = __builtin_readcyclecounter();
  const unsigned long long rt_loop = __builtin_amdgcn_s_memrealtime();
#endif
  int q = 0;
  for (; q + 2 < nq; ++q) chunk(q, T{}, T{});
  if (q + 1 < nq) chunk(q++, T{}, F{});  // last staged chunk: nothing left to load
  chunk(q, F{}, F{});                    // last chunk: MFMAs only
#ifdef DSG_H2_TIMING
  const unsigned long long rt_loop_end = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t_loop_cycles = __builtin_readcyclecounter() - t_begin;
#endif

  // Epilogue.  All global accesses are buffer operations on descriptors that start at this tile's first output
  // channel: the per-lane offset (row, column, +4 channels for the upper half-wave) is one VGPR computed once, the
  // (channel, row) part of each access is a scalar offset, and channels past cout fall outside the descriptor's
  // range -- loads return 0, stores are dropped -- so there is neither address arithmetic nor a bounds branch per
  // element.  The residual values of a 32-channel slab are all in flight before the first use (with one wave per
  // SIMD a load->add->store chain per element would expose the memory latency 64 times).
  // (the host only dispatches here when cout % 8 == 0, so a 4-row half-group is never split by cout)
  const bool has_r = p.res != nullptr;
  const int oscale = GM == 2 ? 2 : 1;  // folded mode: the output map is twice the tiled (low-resolution) grid
  const int oplane = p.hout * p.wout * oscale * oscale;
#ifdef DSG_H2_TIMING_NOSTATS
  const bool want_stats = false;  // (timing experiment: the record buffer is p.stats, the statistics path stays off)
#else
  const bool want_stats = p.stats != nullptr;
#endif
  float* red = reinterpret_cast<float*>(smem_raw);  // [wave][sum | sumsq][cout 64] (the K loop is done with LDS)
  constexpr int RED_FLOATS = NW * (NT / 2) * 2 * BM;
  float* red_lane = (l31 == 16) ? red + 4 * half : red + RED_FLOATS + 64 + lane;  // (+ crel etc. per value)
  const int nvalid = min(BM, p.cout - m0);        // output channels of this tile that exist
  const size_t tile_off = ((size_t)n * p.cout + m0) * oplane;
  const int range = nvalid * oplane * 4;
  const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(p.dst + tile_off, 0, range, 0x00020000);
  const __amdgpu_buffer_rsrc_t res_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_r ? p.res + tile_off : p.dst), 0, has_r ? range : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t bias_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.bias ? p.bias + m0 : p.dst), 0, p.bias ? nvalid * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t temb_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.temb ? p.temb + (size_t)n * p.temb_stride + m0 : p.dst), 0, p.temb ? nvalid * 4 : 0, 0x00020000);
  int voff[NT];  // bytes, per lane and row; the channel part of an address is the scalar offset
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if constexpr (DB) {  // [C/8][H][W][8]: pixel * 32 bytes + this half-wave's four channels
      const int oy = GM == 2 ? 2 * (oy0 + wave * NT + nt) + (phase >> 1) : oy0 + wave * NT + nt;
      const int ox = GM == 2 ? 2 * (ox0 + l31) + (phase & 1) : ox0 + l31;
      voff[nt] = (GM == 2 || ox0 + l31 < p.wout) ? ((oy * (p.wout * oscale) + ox) * 8 + 4 * half) * 4 : 0x7FFFFFF0;
    } else {
      voff[nt] = GM == 2 ? (4 * half * oplane + (2 * (oy0 + wave * NT + nt) + (phase >> 1)) * (2 * p.wout) +
                            2 * (ox0 + l31) + (phase & 1)) * 4
                         : (ox0 + l31 < p.wout ? (4 * half * oplane + (oy0 + wave * NT + nt) * p.wout + ox0 + l31) * 4
                                               : 0x7FFFFFF0);  // (narrow maps: past the last column -> out of range)
    }
  }
  const int oplane4 = __builtin_amdgcn_readfirstlane(oplane * 4);
  // NARROW: maps less than one tile wide (16x16, 8x8): lanes past the last column store nothing (their offset is out
  // of the descriptor's range) and count as zeros in the statistics
  const bool lane_ok = ox0 + l31 < p.wout;
#ifdef DSG_H2_TIMING
  unsigned long long rt_e[6] = {0, 0, 0, 0, 0, 0};
#define DSG_ET(i) do { __builtin_amdgcn_sched_barrier(0); rt_e[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DSG_ET(i)
#endif
  DSG_ET(0);
  auto epilogue = [&](auto stats_tag, auto narrow_tag) {
    constexpr bool STATS = decltype(stats_tag)::value, NARROW = decltype(narrow_tag)::value;
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) {
      float rv[16][NT], addv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int crel = mt * 32 + (r & 3) + 8 * (r >> 2);  // this lane's channel is crel + 4*half
        addv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bias_rs, 16 * half, crel * 4, 0)) +
                  __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(temb_rs, 16 * half, crel * 4, 0));
      }
      if (has_r) {
        if constexpr (DB) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const float4 q = __builtin_bit_cast(
                  float4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, voff[nt], (mt * 4 + rg) * 8 * oplane4, 0));
              rv[4 * rg][nt] = q.x; rv[4 * rg + 1][nt] = q.y; rv[4 * rg + 2][nt] = q.z; rv[4 * rg + 3][nt] = q.w;
            }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int crel = mt * 32 + (r & 3) + 8 * (r >> 2);
              rv[r][nt] = __builtin_bit_cast(
                  float, __builtin_amdgcn_raw_buffer_load_b32(res_rs, voff[nt], crel * oplane4, 0));
            }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) rv[r][nt] = 0.f;
      }
#ifdef DSG_H2_TIMING
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (timing build: the slab's loads have landed)
#endif
      DSG_ET(1 + 2 * mt);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {  // a register group = four consecutive output channels
        float vv[4][NT];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int r = 4 * rg + j;
            vv[j][nt] = ((acc_hi[mt][nt][r] + acc_lo[mt][nt][r] * (1.0f / 2048.0f)) + addv[r]) + rv[r][nt];
          }
        if constexpr (DB) {  // the group is 16 contiguous bytes of the pixel's channel block
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const float4 o = make_float4(vv[0][nt], vv[1][nt], vv[2][nt], vv[3][nt]);
            // The channel-block offset goes into the VECTOR offset on purpose.  With an SGPR soffset hipcc 7.2 treats
            // the 16-byte store as free of the "VALU overwrites store data" hazard and re-uses the data registers
            // two or three instructions later; on gfx950 that corrupted the second dword of lanes 12..15 of every
            // row (found by the bit-exact layout tests).  Without an soffset register it inserts the wait states.
#ifndef DSG_H2_TIMING_NOSTORE
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), dst_rs,
                                                   voff[nt] + (mt * 4 + rg) * 8 * oplane4, 0, 0);
#else
            if (o.x == 1234.5f) red[lane] = o.y + o.z + o.w;  // (timing experiment: keep the math, drop the stores)
#endif
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vv[j][nt]), dst_rs, voff[nt],
                                                    (mt * 32 + j + 8 * rg) * oplane4, 0);
        }
        if (STATS) {  // GroupNorm statistics of the tensor just produced (the next layer's norm reads them)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int crel = mt * 32 + j + 8 * rg;
#pragma unroll
            for (int pr = 0; pr < NT / 2; ++pr) {  // one partial per pair of rows: the same summation tree for any NT
              const float a = (NARROW && !lane_ok) ? 0.f : vv[j][2 * pr], b = (NARROW && !lane_ok) ? 0.f : vv[j][2 * pr + 1];
              const float t1 = half_wave_sum(a + b), t2 = half_wave_sum(a * a + b * b);
              // every lane stores -- lanes 16 / 48 to the real slot, the others to a per-lane dump area behind it:
              // a predicated store here is a branch, and 128 branches fence the scheduler between the DPP chains
              red_lane[((wave * (NT / 2) + pr) * 2 + 0) * BM + crel] = t1;
              red_lane[((wave * (NT / 2) + pr) * 2 + 1) * BM + crel] = t2;
            }
          }
        }
      }
      DSG_ET(2 + 2 * mt);
    }
  };
  if (p.wout < H2_TW) {
    if (want_stats) epilogue(T{}, T{});
    else epilogue(F{}, T{});
  } else {
    if (want_stats) epilogue(T{}, F{});
    else epilogue(F{}, F{});
  }
  if (want_stats) {
    // statistics tiles are 8 rows x 32 columns (4 row pairs, summed in row order in fp64) whatever NT is, so the
    // values -- and everything downstream of the norm -- do not depend on the launch geometry
    __syncthreads();
    if (tid < 2 * BM) {
      const int cl = tid & (BM - 1), which = (tid / BM) & 1;
      if (m0 + cl < p.cout) {
        constexpr int NE = NW * NT / 8;  // 8-row statistics tiles per workgroup tile
        const int ntile1 = p.tiles_x * p.tiles_y * NE;           // entries per phase
        const int ntile = ntile1 * (GM == 2 ? 4 : 1);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          double t = 0.0;
#pragma unroll
          for (int j = 0; j < 4; ++j) t += (double)red[((4 * e + j) * 2 + which) * BM + cl];
          const int tile8 = phase * ntile1 + (ty * NE + e) * p.tiles_x + tx;
#ifndef DSG_H2_TIMING
          p.stats[(((size_t)n * p.cout + m0 + cl) * ntile + tile8) * 2 + which] = t;
#endif
        }
      }
    }
  }
#ifdef DSG_H2_TIMING
  if (p.stats && tid == 0) p.stats[8 + 4 * (size_t)gridDim.x + blockIdx.x] =
      (double)__builtin_amdgcn_s_memrealtime();
#endif
#ifdef DSG_H2_TIMING
  if (p.stats && lane == 0) {
    atomicAdd(&p.stats[0], (double)t_loop_cycles);
    atomicAdd(&p.stats[1], (double)t_vm);
    atomicAdd(&p.stats[2], (double)t_bar);
    atomicAdd(&p.stats[3], 1.0);
    for (int t = 0; t < TAPS; ++t) atomicAdd(&p.stats[8 + 5 * (size_t)gridDim.x + t], (double)t_tap[t]);
    if (wave == 0) {  // per-block record: start / loop begin / loop end (10-ns ticks), loop cycles
      double* rec = p.stats + 8 + 4 * (size_t)blockIdx.x;
      rec[0] = (double)rt_entry;
      rec[1] = (double)rt_loop;
      rec[2] = (double)rt_loop_end;
      rec[3] = (double)t_loop_cycles;
      double* pr = p.stats + 8 + 5 * (size_t)gridDim.x + 16 + 4 * (size_t)blockIdx.x;
      for (int k = 0; k < 4; ++k) pr[k] = (double)rt_p[k];
      double* er = p.stats + 8 + 9 * (size_t)gridDim.x + 16 + 6 * (size_t)blockIdx.x;
      rt_e[5] = __builtin_amdgcn_s_memrealtime();
      for (int k = 0; k < 6; ++k) er[k] = (double)rt_e[k];
    }
  }
#endif
}

// OIHW fp32 -> [cin/16][piece 2][tap k*k][g 2][cout_pad][8] fp16 (hi, scaled lo); zero-padded couts.
// mode 1 (data gradient): the conv dX = conv(dY, W^T flipped) has K = cout, N = cin: the same layout with the
// roles of the two channel axes swapped and the taps reversed.  `cout_off` places this weight's N columns inside a
// wider matrix (fused q/k/v projection).
__global__ void weight_relayout_h2_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int cout_w, int cin_w,
                                          int taps, int cout_pad, int cout_off, int mode) {
  const int cin = mode ? cout_w : cin_w;   // K axis of the conv this layout feeds
  const int cout = mode ? cin_w : cout_w;  // N axis (this weight's share of it)
  const int64_t total = (int64_t)(cin / 16) * taps * 2 * cout * 8;  // one thread per (chunk, tap, g, co, j)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % 8);
    int64_t r = i / 8;
    const int co = (int)(r % cout);
    r /= cout;
    const int g = (int)(r % 2);
    r /= 2;
    const int tap = (int)(r % taps);
    const int q = (int)(r / taps);
    const int ci = q * 16 + g * 8 + j;
    const float v = mode ? w[((int64_t)ci * cin_w + co) * taps + (taps - 1 - tap)] : w[((int64_t)co * cin_w + ci) * taps + tap];
    const _Float16 h1 = (_Float16)v;
    const _Float16 h2 = (_Float16)((v - (float)h1) * 2048.0f);
    const int64_t base = ((((int64_t)q * 2 + 0) * taps + tap) * 2 + g) * cout_pad + cout_off + co;
    const int64_t base1 = ((((int64_t)q * 2 + 1) * taps + tap) * 2 + g) * cout_pad + cout_off + co;
    dst[base * 8 + j] = h1;
    dst[base1 * 8 + j] = h2;
  }
}

static int g_h2_enabled = 1;
static int g_h2_waves = 4;  // 16-row tiles: 4 waves x 4 rows or 8 waves x 2 rows (tuning key 6)
static int g_h2_fold = 1;   // folded up-sampler convs (tuning key 8: A/B against the x2 gather)
static int g_h2_stats = 1;  // epilogue GroupNorm statistics (tuning key 5: A/B against the separate pass)
static int g_h2_bm32 = 0;      // 32-cout x 8-row workgroups, two per CU, for the shallow levels (tuning key 16; measured
                               // slower than the 64 x 16 geometry: 0.356 vs 0.302 ms at 64 channels / 256^2 -- off)
static int g_h2_bm32_min = 512;  // ... when the 64-cout x 16-row grid has at least this many workgroups
static int g_h2_bm32_small = 1;  // 32-cout workgroups for grids of at most half the CUs (tuning key 17)
constexpr int H2_CUS = 256;
static int g_h2_s2 = 1;  // stride-2 convs on the split path (tuning key 15: A/B against the f32 MFMA kernel)
static int g_h2_pw_occ2 = 1;  // pointwise convs: 8-row tiles compiled for two workgroups per CU (tuning key 11)
static int g_h2_rows = 0;  // rows per wave: 0 = by grid size, 2 | 4 forced (tuning key 3)

// folded up-sampler mode: nearest x2 + 3x3 as four 2x2 convs of the low-resolution input (weight_h2_fold)
static bool conv_h2_fold(const dsg_conv_args* a) {
  return g_h2_enabled && g_h2_fold && a->weight_h2_fold != nullptr && a->upsample == 1 && a->ksize == 3 && a->stride == 1 &&
         !a->pool2 && !a->gn_scale_shift && !a->weight_h2_cout_stride && (a->c0 + a->c1) % 16 == 0 &&
         (a->c1 == 0 || a->c0 % 16 == 0) && a->win % H2_TW == 0 && a->hin % 8 == 0 && a->cout % 8 == 0 &&
         (a->c0 + a->c1) <= 1024;
}

// stride-2 3x3 conv as a 2x2 conv over the space-to-depth image (GM = 3): channel-blocked tensors only
static bool conv_h2_s2(const dsg_conv_args* a, int hout, int wout) {
  return g_h2_enabled && g_h2_s2 && a->weight_h2_s2 != nullptr && a->stride == 2 && a->ksize == 3 && !a->upsample &&
         !a->pool2 && !a->gn_scale_shift && a->src_layout == 1 && a->dst_layout == 1 && a->c1 == 0 && a->c0 % 8 == 0 &&
         a->cout % 8 == 0 && a->hin % 2 == 0 && a->win % 2 == 0 && hout % 8 == 0 && wout % H2_TW == 0;
}

bool conv_h2_eligible(const dsg_conv_args* a, int hout, int wout) {
  const int cin = a->c0 + a->c1;
  if (conv_h2_s2(a, hout, wout)) return true;
  if (a->stride != 1) return false;
  const int lay = (a->src_layout ? 1 : 0) | (a->dst_layout ? 2 : 0);
  // channel-blocked tensors: 3x3 stride-1 convs (plain or folded up-sampler) with every tensor blocked; pointwise
  // convs with any pair of layouts
  if (lay && a->ksize == 3 && (lay != 3 || (a->upsample && !conv_h2_fold(a)))) return false;
  if (lay && (a->c0 % 8 || a->c1 % 8 || a->cout % 8)) return false;
  if (conv_h2_fold(a)) return true;
  if (a->weight_h2_cout_stride && (a->weight_h2_cout_stride % 64 || a->weight_h2_cout_stride < (a->cout + 63) / 64 * 64))
    return false;
  if (cin > 1024) return false;  // the GroupNorm scale/shift table shares LDS with the K-chunk buffers
  if (a->ksize == 1)  // pointwise: the map is re-tiled as (h*w/32) rows of 32 pixels, so only h*w matters
    return g_h2_enabled && a->weight_h2 != nullptr && a->stride == 1 && !a->upsample && !a->pool2 && !a->temb &&
           cin % 16 == 0 && (a->c1 == 0 || a->c0 % 16 == 0) && (hout * wout) % (8 * H2_TW) == 0 && a->cout % 8 == 0;
  if (a->gn_scale_shift && (!a->silu || a->upsample)) return false;  // combinations the U-Net does not have
  return g_h2_enabled && a->weight_h2 != nullptr && a->stride == 1 && a->upsample <= 1 && !a->pool2 &&
         cin % 16 == 0 && (a->c1 == 0 || a->c0 % 16 == 0) && (wout % H2_TW == 0 || wout == 16 || wout == 8) && (hout % 8 == 0) &&
         a->cout % 8 == 0;
}

// tile geometry shared by the launcher and dsg_conv2d_stats_tiles
// 16-row tiles (4 rows per wave) are the efficient shape; 8-row tiles double the workgroup count.  The chip runs
// 256 workgroups at a time, so what counts is the number of ROUNDS: an 8-row workgroup costs ~0.55 of a 16-row one
// (half the MFMAs, the same fixed cost), and e.g. 320 workgroups of 16 rows (2 rounds) lose to 640 of 8 (3 x 0.55).
static bool rows16_pays(int b16) {
  if (g_h2_rows == 2 || b16 <= 0) return false;
  if (g_h2_rows == 4) return true;
  if (b16 < 256) return false;
  const int r16 = (b16 + 255) / 256, r8 = (2 * b16 + 255) / 256;
  return 100 * r16 <= 55 * r8;
}

static bool conv_h2_rows16(const dsg_conv_args* a, int hout, int wout) {
  if (conv_h2_fold(a)) {  // tiled on the low-resolution grid, four phases per cout tile
    const int cp = (a->cout + 63) / 64 * 64;
    return rows16_pays((a->hin % 16 == 0) ? (a->win / H2_TW) * (a->hin / 16) * a->n * (cp / H2_BM) * 4 : 0);
  }
  if (a->ksize == 1) {
    hout = hout * wout / H2_TW;
    wout = H2_TW;
  }
  const int cout_pad = (a->cout + 63) / 64 * 64;
  return rows16_pays((hout % 16 == 0) ? (wout / H2_TW) * (hout / 16) * a->n * (cout_pad / H2_BM) : 0);
}

int conv_h2_stats_tiles(const dsg_conv_args* a, int hout, int wout) {
  if (!g_h2_stats || !conv_h2_eligible(a, hout, wout)) return 0;
  if (a->ksize == 1) return hout * wout / (8 * H2_TW);  // (pointwise: the map is re-tiled as rows of 32 pixels)
  return (hout / 8) * ((wout + H2_TW - 1) / H2_TW);  // 8-row x 32-column statistics tiles for either block height
}

template <int GM, int NT, int KS, int ACT, int NW = 4, int OCC = 1, int LAY = 0, int BM = 64>
static int h2_launch(dim3 grid, size_t lds, hipStream_t st, const ConvH2P& p) {
  auto kern = conv_h2_kernel<GM, NT, KS, ACT, NW, OCC, LAY, BM>;
  static bool raised = false;
  if (!raised) {
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    raised = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, p);
  return DSG_OK;
}

int conv_h2_launch(const dsg_conv_args* a, int hout, int wout, hipStream_t st) {
  // pointwise: 8-row tiles, two workgroups per CU (the only pointwise kernels that take channel-blocked tensors)
  const bool occ2 = a->ksize == 1 && (g_h2_pw_occ2 || a->src_layout || a->dst_layout);
  const bool nt4 = !occ2 && conv_h2_rows16(a, hout, wout);
  ConvH2P p;
  p.stats = a->stats_out;
  p.src0 = a->src0; p.src1 = a->src1; p.c0 = a->c0; p.c1 = a->c1; p.cin = a->c0 + a->c1;
  p.n = a->n; p.hin = a->hin; p.win = a->win;
  if (a->ksize == 1) {
    hout = hout * wout / H2_TW; wout = H2_TW;
    p.hin = hout; p.win = wout;
  }
  const bool fold = conv_h2_fold(a);
  const bool s2 = conv_h2_s2(a, hout, wout);
  if (fold) {  // the kernel tiles the LOW-resolution grid; outputs land at (2y + py, 2x + px)
    hout = a->hin;
    wout = a->win;
  }
  p.hc = (a->upsample && !fold) ? 2 * p.hin : p.hin;
  p.wc = (a->upsample && !fold) ? 2 * p.win : p.win;
  if (s2) {  // the patch lives on the output grid; its 4 C "channels" are (channel block, pixel parity) groups
    p.hc = hout;
    p.wc = wout;
    p.cin = 4 * a->c0;
  }
  p.hout = hout; p.wout = wout; p.cout = a->cout; p.cout_pad = (a->cout + 63) / 64 * 64;
  p.wh_stride = a->weight_h2_cout_stride ? a->weight_h2_cout_stride : p.cout_pad;
  p.wh = static_cast<const _Float16*>(fold ? a->weight_h2_fold : (s2 ? a->weight_h2_s2 : a->weight_h2));
  p.bias = a->bias; p.ss = a->gn_scale_shift; p.silu = a->silu; p.temb = a->temb; p.temb_stride = a->temb_stride;
  p.res = a->residual; p.dst = a->dst;
  // 16-row tiles (NT = 4) when they still give every CU a workgroup; 8-row tiles otherwise
  const int th = nt4 ? 16 : 8;
  p.tiles_x = (wout + H2_TW - 1) / H2_TW; p.tiles_y = hout / th;
  const bool k1 = a->ksize == 1;
  const size_t lds = 2 * (size_t)(k1 ? (nt4 ? H2Geom<4, 1>::BUF_BYTES : H2Geom<2, 1>::BUF_BYTES)
                                     : (nt4 ? H2Geom<4, 3>::BUF_BYTES : H2Geom<2, 3>::BUF_BYTES)) +
                     (a->gn_scale_shift ? (size_t)p.cin * 2 * sizeof(float) : 0);  // + the scale/shift table
  dim3 grid(p.tiles_x * p.tiles_y * p.n * (p.cout_pad / H2_BM) * (fold ? 4 : 1));
  int pi = -1;
  if (prof_on()) {
    const double px = (double)p.n * hout * wout * (fold ? 4 : 1);  // output pixels; FLOPs are the reference op's
    const int taps = a->ksize * a->ksize;
    const double cin_ref = a->c0 + a->c1;  // (stride 2: the kernel's 4 C x 4 taps are the reference op's C x 9)
    pi = prof_begin(s2 ? 2 : (a->ksize == 1 ? 8 : (a->upsample ? 7 : 6)), 2.0 * px * p.cout * cin_ref * taps,
                    4.0 * ((double)p.n * cin_ref * p.hin * p.win + cin_ref * taps * p.cout +
                           px * p.cout * (p.res ? 2.0 : 1.0)), st);
  }
  const int act = a->gn_scale_shift ? (a->silu ? 2 : 3) : 0;
  const int lay = (a->src_layout ? 1 : 0) | (a->dst_layout ? 2 : 0);
  int rc = DSG_OK;
#define DSG_H2_LAUNCH(GM, KS, ACT)                                                  \
  do {                                                                              \
    if (nt4 && g_h2_waves == 8) rc = h2_launch<GM, 2, KS, ACT, 8>(grid, lds, st, p); \
    else if (nt4) rc = h2_launch<GM, 4, KS, ACT>(grid, lds, st, p);                 \
    else rc = h2_launch<GM, 2, KS, ACT>(grid, lds, st, p);                          \
  } while (0)
#define DSG_H2_LAUNCH_BLK(GM, KS, ACT) /* every tensor channel-blocked: four-wave kernels only */ \
  do {                                                                              \
    if (nt4) rc = h2_launch<GM, 4, KS, ACT, 4, 1, 3>(grid, lds, st, p);             \
    else rc = h2_launch<GM, 2, KS, ACT, 4, 1, 3>(grid, lds, st, p);                 \
  } while (0)
#define DSG_H2_LAUNCH_PW(ACT) /* pointwise, two workgroups per CU: any layout pair */ \
  do {                                                                              \
    if (lay == 0) rc = h2_launch<0, 2, 1, ACT, 4, 2, 0>(grid, lds, st, p);          \
    else if (lay == 1) rc = h2_launch<0, 2, 1, ACT, 4, 2, 1>(grid, lds, st, p);     \
    else if (lay == 2) rc = h2_launch<0, 2, 1, ACT, 4, 2, 2>(grid, lds, st, p);     \
    else rc = h2_launch<0, 2, 1, ACT, 4, 2, 3>(grid, lds, st, p);                   \
  } while (0)
  // 32-cout workgroups: (a) optional, two per CU on the shallow levels (cin <= 128: LDS); (b) small batches: when
  // even the 8-row x 64-cout grid leaves more than half of the CUs idle, halve the cout tile to double the grid
  const bool bm32_ok = lay == 3 && !fold && !s2 && !k1 && !a->upsample && wout % H2_TW == 0;
  const bool bm32 = bm32_ok && ((g_h2_bm32 && p.cin <= 128 && (int)grid.x >= g_h2_bm32_min) ||
                                (g_h2_bm32_small && !nt4 && (int)grid.x <= H2_CUS / 2));
  if (s2) {
    DSG_H2_LAUNCH_BLK(3, 3, 0);
  } else if (fold) {
    if (lay) DSG_H2_LAUNCH_BLK(2, 3, 0);
    else if (nt4) rc = h2_launch<2, 4, 3, 0>(grid, lds, st, p);
    else rc = h2_launch<2, 2, 3, 0>(grid, lds, st, p);
  } else if (k1 && occ2) {
    if (act == 0) DSG_H2_LAUNCH_PW(0);
    else DSG_H2_LAUNCH_PW(3);
  } else if (k1) {
    if (act == 0) DSG_H2_LAUNCH(0, 1, 0);
    else DSG_H2_LAUNCH(0, 1, 3);
  } else if (a->upsample) {
    DSG_H2_LAUNCH(1, 3, 0);
  } else if (bm32) {
    // shallow levels: 32 couts x 8 rows x 32 columns per workgroup, two workgroups per CU
    const dim3 g32(((wout + H2_TW - 1) / H2_TW) * (hout / 8) * p.n * (p.cout_pad / 32));
    const size_t lds32 = 2 * (size_t)H2Geom<2, 3, 4, 9, 32>::BUF_BYTES + (a->gn_scale_shift ? (size_t)p.cin * 2 * sizeof(float) : 0);
    ConvH2P q = p;
    q.tiles_y = hout / 8;
    if (act == 0) rc = h2_launch<0, 2, 3, 0, 4, 2, 3, 32>(g32, lds32, st, q);
    else rc = h2_launch<0, 2, 3, 2, 4, 2, 3, 32>(g32, lds32, st, q);
  } else if (lay) {
    if (act == 0) DSG_H2_LAUNCH_BLK(0, 3, 0);
    else DSG_H2_LAUNCH_BLK(0, 3, 2);
  } else {
    if (act == 0) DSG_H2_LAUNCH(0, 3, 0);
    else DSG_H2_LAUNCH(0, 3, 2);
  }
#undef DSG_H2_LAUNCH
#undef DSG_H2_LAUNCH_BLK
#undef DSG_H2_LAUNCH_PW
  if (rc != DSG_OK) return rc;
  prof_end(pi, st);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

void conv_h2_set_enabled(int on) { g_h2_enabled = on; }
void conv_h2_set_rows(int r) { g_h2_rows = r; }
void conv_h2_set_stats(int on) { g_h2_stats = on; }
void conv_h2_set_fold(int on) { g_h2_fold = on; }
void conv_h2_set_waves(int w) { g_h2_waves = w; }
void conv_h2_set_pw_occ2(int v) { g_h2_pw_occ2 = v; }
void conv_h2_set_s2(int v) { g_h2_s2 = v; }
void conv_h2_set_bm32_small(int v) { g_h2_bm32_small = v; }
void conv_h2_set_bm32(int v) { g_h2_bm32 = v != 0; if (v > 1) g_h2_bm32_min = v; }

}  // namespace dsg

// OIHW 3x3 -> folded up-sampler weights [phase 4][cin/16][piece 2][tap 2x2][g 2][cout_pad][8]: for output phase
// (py, px) the 2x2 taps are sums of the 3x3 taps that land on the same low-resolution pixel -- rows {0 | 1+2} for
// py = 0, {0+1 | 2} for py = 1, the same in x (summed in fp32, dy outer / dx inner, then split hi / lo).
__global__ void weight_fold_h2_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int cout, int cin,
                                      int cout_pad) {
  const int64_t per_phase = (int64_t)(cin / 16) * 4 * 2 * cout * 8;
  const int64_t total = 4 * per_phase;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % 8);
    int64_t r = i / 8;
    const int co = (int)(r % cout);
    r /= cout;
    const int g = (int)(r % 2);
    r /= 2;
    const int tap = (int)(r % 4);
    r /= 4;
    const int q = (int)(r % (cin / 16));
    const int phase = (int)(r / (cin / 16));
    const int py = phase >> 1, px = phase & 1, tr = tap >> 1, tc = tap & 1;
    const int dy0 = py == 0 ? (tr == 0 ? 0 : 1) : (tr == 0 ? 0 : 2), dy1 = py == 0 ? (tr == 0 ? 0 : 2) : (tr == 0 ? 1 : 2);
    const int dx0 = px == 0 ? (tc == 0 ? 0 : 1) : (tc == 0 ? 0 : 2), dx1 = px == 0 ? (tc == 0 ? 0 : 2) : (tc == 0 ? 1 : 2);
    const int ci = q * 16 + g * 8 + j;
    const float* wp = w + ((int64_t)co * cin + ci) * 9;
    float v = 0.f;
    for (int dy = dy0; dy <= dy1; ++dy)
      for (int dx = dx0; dx <= dx1; ++dx) v += wp[dy * 3 + dx];
    const _Float16 h1 = (_Float16)v;
    const _Float16 h2 = (_Float16)((v - (float)h1) * 2048.0f);
    const int64_t pb = (int64_t)phase * (cin / 16) + q;
    dst[((((pb * 2 + 0) * 4 + tap) * 2 + g) * cout_pad + co) * 8 + j] = h1;
    dst[((((pb * 2 + 1) * 4 + tap) * 2 + g) * cout_pad + co) * 8 + j] = h2;
  }
}

// OIHW 3x3 -> stride-2 weights over the space-to-depth image [4 cin/16][piece 2][tap 2x2][g 2][cout_pad][8]:
// k' = (channel block cb, parity (py, px), channel j of the block); tap (ty, tx) of parity (py, px) is the 3x3 tap
// (2 ty + py - 1, 2 tx + px - 1) where that exists, zero where it does not.
__global__ void weight_s2_h2_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int cout, int cin,
                                    int cout_pad) {
  const int nq = 4 * cin / 16;
  const int64_t total = (int64_t)nq * 4 * 2 * cout * 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % 8);
    int64_t r = i / 8;
    const int co = (int)(r % cout);
    r /= cout;
    const int g = (int)(r % 2);
    r /= 2;
    const int tap = (int)(r % 4);
    const int q = (int)(r / 4);
    const int gi = 2 * q + g, cb = gi >> 2, pp = gi & 3;
    const int dy = 2 * (tap >> 1) + (pp >> 1) - 1, dx = 2 * (tap & 1) + (pp & 1) - 1;
    const float v = (dy >= 0 && dx >= 0) ? w[((int64_t)co * cin + cb * 8 + j) * 9 + dy * 3 + dx] : 0.f;  // (dy, dx <= 2)
    const _Float16 h1 = (_Float16)v;
    const _Float16 h2 = (_Float16)((v - (float)h1) * 2048.0f);
    dst[(((((int64_t)q * 2 + 0) * 4 + tap) * 2 + g) * cout_pad + co) * 8 + j] = h1;
    dst[(((((int64_t)q * 2 + 1) * 4 + tap) * 2 + g) * cout_pad + co) * 8 + j] = h2;
  }
}

static int relayout_h2(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, int32_t ksize, int32_t n_total,
                       int32_t n_off, int mode, void* stream) {
  DSG_CHECK_ARG(w_oihw && dst_half, "dsg_conv_weight_relayout_h2: NULL pointer");
  DSG_CHECK_ARG(ksize == 1 || ksize == 3, "dsg_conv_weight_relayout_h2: ksize must be 1 or 3");
  const int k_axis = mode ? cout : cin, n_axis = mode ? cin : cout;
  DSG_CHECK_ARG(cout > 0 && cin > 0 && k_axis % 16 == 0,
                "dsg_conv_weight_relayout_h2: the contraction axis (%d) must be a positive multiple of 16", k_axis);
  if (n_total == 0) n_total = n_axis;
  DSG_CHECK_ARG(n_off >= 0 && n_off + n_axis <= n_total, "dsg_conv_weight_relayout_h2: column window out of range");
  const int cout_pad = (n_total + 63) / 64 * 64;
  const int taps = ksize * ksize;
  const int64_t total = (int64_t)(k_axis / 16) * taps * 2 * n_axis * 8;
  const int blocks = (int)std::min<int64_t>(dsg::cdiv64(total, 256), 4096);
  hipLaunchKernelGGL(dsg::weight_relayout_h2_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w_oihw, static_cast<_Float16*>(dst_half), cout, cin, taps, cout_pad, n_off, mode);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_conv_weight_relayout_h2(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, int32_t ksize,
                                        int32_t cout_total, int32_t cout_off, void* stream) {
  return relayout_h2(w_oihw, dst_half, cout, cin, ksize, cout_total, cout_off, 0, stream);
}

DSG_API int dsg_conv_weight_relayout_h2_fold(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, void* stream) {
  DSG_CHECK_ARG(w_oihw && dst_half, "dsg_conv_weight_relayout_h2_fold: NULL pointer");
  DSG_CHECK_ARG(cout > 0 && cin > 0 && cin % 16 == 0, "dsg_conv_weight_relayout_h2_fold: cin (%d) must be a positive multiple of 16", cin);
  const int cout_pad = (cout + 63) / 64 * 64;
  const int64_t total = (int64_t)4 * (cin / 16) * 4 * 2 * cout * 8;
  const int blocks = (int)std::min<int64_t>(dsg::cdiv64(total, 256), 4096);
  hipLaunchKernelGGL(weight_fold_h2_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw,
                     static_cast<_Float16*>(dst_half), cout, cin, cout_pad);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_conv_weight_relayout_h2_s2(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin, void* stream) {
  DSG_CHECK_ARG(w_oihw && dst_half, "dsg_conv_weight_relayout_h2_s2: NULL pointer");
  DSG_CHECK_ARG(cout > 0 && cin > 0 && cin % 8 == 0, "dsg_conv_weight_relayout_h2_s2: cin (%d) must be a positive multiple of 8", cin);
  const int cout_pad = (cout + 63) / 64 * 64;
  const int64_t total = (int64_t)(4 * cin / 16) * 4 * 2 * cout * 8;
  const int blocks = (int)std::min<int64_t>(dsg::cdiv64(total, 256), 4096);
  hipLaunchKernelGGL(weight_s2_h2_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw,
                     static_cast<_Float16*>(dst_half), cout, cin, cout_pad);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_conv_weight_relayout_h2_dgrad(const float* w_oihw, void* dst_half, int32_t cout, int32_t cin,
                                              int32_t ksize, void* stream) {
  return relayout_h2(w_oihw, dst_half, cout, cin, ksize, 0, 0, 1, stream);
}
