// Self-attention core softmax(Q^T K / sqrt(d)) V for gfx950, fp32, head_dim 8..64.
//
// Replaces F.scaled_dot_product_attention inside diffusers' Attention block as instantiated by
// UNetMidBlock2D (and Attn{Down,Up}Block2D) of the UNet2DModel that DriveSceneGen builds at
// DriveSceneGen/scripts/train.py:39-57 (attention_head_dim = 8 -> 64 heads x 8 dims over 1024 tokens;
// SURVEY.md App. A.2).  0.6 % of the network's FLOPs; K = d = 8 starves an MFMA tile, so this first
// version runs the two tiny contractions on the VALU with K/V tiles broadcast from LDS and a
// chunked online softmax; each thread owns QPT query rows so a K/V LDS read is amortised QPT times.
//
// Layout: qkv [N][3C][L] is the output of the fused q/k/v projection (a 1x1 conv over [N,C,L]);
// channel = head*D + i, so a head's q/k/v are D rows of L contiguous floats.  out is [N][C][L].
#include "dsg_h16.h"

namespace dsg {

constexpr int ATT_LDS_FLOATS = 4096;  // per K and per V tile: 16 KiB each, keys per tile = 4096 / D
constexpr int ATT_KB = 8;    // keys per online-softmax chunk

template <int D, int QPT>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        float* __restrict__ lse, int c, int heads, int l, float qscale) {
  constexpr int ATT_KT = ATT_LDS_FLOATS / D;
  __shared__ __attribute__((aligned(16))) float KVl[2 * ATT_KT * D];
  float* Kl = KVl;
  float* Vl = KVl + ATT_KT * D;
  const int tid = threadIdx.x;
  const int h = blockIdx.y, n = blockIdx.z;
  const float* qp = qkv + ((size_t)n * 3 * c + h * D) * l;
  const float* kp = qp + (size_t)c * l;
  const float* vp = kp + (size_t)c * l;

  float q[QPT][D], o[QPT][D], m[QPT], lsum[QPT];
  int qi[QPT];
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    qi[u] = (blockIdx.x * QPT + u) * 256 + tid;
    const int qc = min(qi[u], l - 1);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      q[u][i] = qp[(size_t)i * l + qc] * qscale;
      o[u][i] = 0.f;
    }
    m[u] = -1e30f;
    lsum[u] = 0.f;
  }

  for (int j0 = 0; j0 < l; j0 += ATT_KT) {
    const int kt = min(ATT_KT, l - j0);
    __syncthreads();
    for (int e = tid; e < ATT_KT * D; e += 256) {
      const int i = e / ATT_KT, j = e - i * ATT_KT;  // coalesced global read along j
      float kv = 0.f, vv = 0.f;
      if (j < kt) {
        kv = kp[(size_t)i * l + j0 + j];
        vv = vp[(size_t)i * l + j0 + j];
      }
      Kl[j * D + i] = kv;
      Vl[j * D + i] = vv;
    }
    __syncthreads();
    const int nchunk = (kt + ATT_KB - 1) / ATT_KB;
    for (int ch = 0; ch < nchunk; ++ch) {
      const int jb = ch * ATT_KB;
      float s[QPT][ATT_KB];
#pragma unroll
      for (int jj = 0; jj < ATT_KB; ++jj) {
        float kr[D];
#pragma unroll
        for (int i4 = 0; i4 < D; i4 += 4) {
          const float4 t = *reinterpret_cast<const float4*>(&Kl[(jb + jj) * D + i4]);
          kr[i4] = t.x; kr[i4 + 1] = t.y; kr[i4 + 2] = t.z; kr[i4 + 3] = t.w;
        }
        const bool valid = (jb + jj) < kt;
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < D; ++i) a = fmaf(q[u][i], kr[i], a);
          s[u][jj] = valid ? a : -1e30f;
        }
      }
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        float mx = s[u][0];
#pragma unroll
        for (int jj = 1; jj < ATT_KB; ++jj) mx = fmaxf(mx, s[u][jj]);
        const float mn = fmaxf(m[u], mx);
        const float sc = exp2f(m[u] - mn);
        m[u] = mn;
        float ps = 0.f;
#pragma unroll
        for (int jj = 0; jj < ATT_KB; ++jj) {
          s[u][jj] = exp2f(s[u][jj] - mn);
          ps += s[u][jj];
        }
        lsum[u] = lsum[u] * sc + ps;
#pragma unroll
        for (int i = 0; i < D; ++i) o[u][i] *= sc;
      }
#pragma unroll
      for (int jj = 0; jj < ATT_KB; ++jj) {
        float vr[D];
#pragma unroll
        for (int i4 = 0; i4 < D; i4 += 4) {
          const float4 t = *reinterpret_cast<const float4*>(&Vl[(jb + jj) * D + i4]);
          vr[i4] = t.x; vr[i4 + 1] = t.y; vr[i4 + 2] = t.z; vr[i4 + 3] = t.w;
        }
#pragma unroll
        for (int u = 0; u < QPT; ++u)
#pragma unroll
          for (int i = 0; i < D; ++i) o[u][i] = fmaf(s[u][jj], vr[i], o[u][i]);
      }
    }
  }

  float* op = out + ((size_t)n * c + h * D) * l;
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    if (qi[u] < l) {
      const float inv = 1.0f / lsum[u];
#pragma unroll
      for (int i = 0; i < D; ++i) op[(size_t)i * l + qi[u]] = o[u][i] * inv;
      // log2-domain log-sum-exp of the scaled scores (training saves it for the backward recompute)
      if (lse) lse[((size_t)n * heads + h) * l + qi[u]] = m[u] + log2f(lsum[u]);
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// head_dim 8 on the matrix cores (the mid-block / Attn*Block2D case of the reference nets: 64 heads x 8).
// K = d = 8 is exactly one v_mfma_f32_32x32x8_f16 step, and the C/D layout of the 32x32 tile (lane = column,
// register r = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) is, for r = 4b .. 4b + 3, exactly the B-operand layout of
// the next 8-deep step.  So with S^T = K^T Q (rows = keys, columns = queries):
//   * a lane owns ONE query and 16 of the tile's 32 keys: the softmax is per lane, one exchange with lane ^ 32;
//   * P^T stays in registers and feeds O = V P as B operands (two 16-deep steps, V read in the matching key order)
//     -- no shuffles, no LDS round trip;
//   * V is the A operand with rows = the 8 head dims, a row of ONES (row 8 of the tile) accumulates the softmax
//     denominator of exactly the P that was multiplied, the other rows are padding.
// fp32-class accuracy as in conv_h2.hip: q, k, v and p are split x = hi + lo * 2^-11 (two fp16 parts), three MFMAs
// per product, the 2^11-scaled cross terms in a second accumulator.
// Workgroup = 8 waves x 32 queries of one (image, head); its K and V (fp16 pairs) sit in LDS, KT keys at a time.
// ---------------------------------------------------------------------------------------------------
typedef _Float16 att_half4 __attribute__((ext_vector_type(4)));
typedef _Float16 att_half8 __attribute__((ext_vector_type(8)));
typedef short att_short4 __attribute__((ext_vector_type(4)));
constexpr int ATM_KT = 512;                  // keys per LDS tile (35 KB of LDS: four workgroups per CU)
constexpr int ATM_VSTR = ATM_KT + 4;         // V row stride in halfs (+8 bytes: rows fall into different banks)

constexpr int ATM_NW = 8;                    // waves per workgroup: 256 queries share one conversion of K and V
#ifndef DSG_ATM_MINW
#define DSG_ATM_MINW 4
#endif

// fp32 -> the 16-bit operand type of PREC, carried in a _Float16-typed container (bits only)
template <int PREC>
__device__ __forceinline__ _Float16 att_cvt(float v) {
  if constexpr (PREC == 1) return __builtin_bit_cast(_Float16, (__bf16)v);
  else return (_Float16)v;
}
// 8-deep S^T = K^T Q step on the matrix cores
template <int PREC>
__device__ __forceinline__ f32x16 att_mma8(att_half4 a, att_half4 b, f32x16 c) {
  if constexpr (PREC == 1)
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(att_short4, a), __builtin_bit_cast(att_short4, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0);
}

// PREC 0: fp32-class accuracy (fp16 pairs, three MFMAs per product).  PREC 1 / 2 (the mixed-precision modes,
// BASELINE.json configs[4] "MFMA bf16 attn"): q, k, v and the probabilities are rounded once to bf16 / fp16, ONE MFMA per
// product, fp32 scores / running maximum / accumulators; the row of ones in V still accumulates the denominator of
// exactly the (rounded) probabilities that were multiplied, so each output row is a convex combination of V rows.
// BLK: q, k, v and the output are CHANNEL-BLOCKED, [N][3C/8][L][8] / [N][C/8][L][8] in the plan's element type (fp32, or
// the 16-bit type of PREC): head_dim 8 is exactly one channel block, so a token's q / k / v vector is one 32- / 16-byte
// piece -- the projection in front writes 16-byte pieces instead of scattering 4-byte words over 3C planes (its kernel
// ran at 0.8-1.2 TB/s), the K / V tiles load as whole vectors, and the out-projection behind reads blocked sources.
template <int PREC, bool BLK = false>
__global__ __launch_bounds__(64 * ATM_NW, DSG_ATM_MINW) void attention_mfma8_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 float* __restrict__ lse, int c, int heads, int l,
                                                                 float qscale) {
  constexpr bool SPLIT = PREC == 0;
  constexpr bool E16 = BLK && PREC != 0;        // 16-bit elements
  constexpr int ESZ = E16 ? 2 : 4;
  typedef unsigned att_u2 __attribute__((ext_vector_type(2)));
  typedef unsigned att_u4 __attribute__((ext_vector_type(4)));
  // 8 consecutive elements of a blocked tensor as fp32
  auto load8 = [&](const char* ptr, float (&v)[8]) {
    if constexpr (E16) {
      const att_u4 w = *reinterpret_cast<const att_u4*>(ptr);
      v[0] = lo16<PREC>(w.x); v[1] = hi16<PREC>(w.x); v[2] = lo16<PREC>(w.y); v[3] = hi16<PREC>(w.y);
      v[4] = lo16<PREC>(w.z); v[5] = hi16<PREC>(w.z); v[6] = lo16<PREC>(w.w); v[7] = hi16<PREC>(w.w);
    } else {
      const float4 a = *reinterpret_cast<const float4*>(ptr), b = *reinterpret_cast<const float4*>(ptr + 16);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
  };
  __shared__ __attribute__((aligned(16))) _Float16 Kh[ATM_KT * 8], Kl[SPLIT ? ATM_KT * 8 : 8];          // [key][d]
  __shared__ __attribute__((aligned(16))) _Float16 Vh[9 * ATM_VSTR], Vl[SPLIT ? 9 * ATM_VSTR : 8];      // [d | ones][key]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  // Workgroup id -> (query tile, head, image).  Consecutive ids go to the 8 XCDs in turn, each with its own L2: XCD k
  // takes a CONTIGUOUS eighth of the (image, head, tile) list, so the query tiles of one head -- which all stream the
  // same K and V -- run behind one L2 at about the same time, and K / V come from HBM once instead of once per XCD
  // (the tile-major order spread a head's 8 tiles over the 8 XCDs: 4.1x the algorithmic traffic, profiles/r01*).
  const int qtiles = (l + 32 * ATM_NW - 1) / (32 * ATM_NW);
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int qt = bid % qtiles, hn = bid / qtiles;
  const int h = hn % heads, n = hn / heads;
  const float* qp = qkv + ((size_t)n * 3 * c + h * 8) * l;
  const float* kp = qp + (size_t)c * l;
  const float* vp = kp + (size_t)c * l;
  // blocked: channel block h of the q / k / v thirds of image n, [L][8] elements each
  const char* qb = reinterpret_cast<const char*>(qkv) + ((size_t)n * 3 * c + h * 8) * l * ESZ;
  const char* kb = qb + (size_t)c * l * ESZ;
  const char* vb = kb + (size_t)c * l * ESZ;
  const int q0 = (qt * ATM_NW + wave) * 32;  // this wave's 32 queries (l % 32 == 0; a wave past the end idles)
  const bool active = q0 < l;
  const int qi = min(q0 + l31, l - 1);

  // B operand of S^T = K^T Q: lane (query l31, half) holds d = 4 half .. 4 half + 3, pre-scaled into the log2 domain
  att_half4 qh, ql;
  float qv8[8];
  if constexpr (BLK) load8(qb + (size_t)qi * 8 * ESZ, qv8);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = (BLK ? (half ? qv8[4 + i] : qv8[i]) : qp[(size_t)(4 * half + i) * l + qi]) * qscale;
    if constexpr (SPLIT) {
      const _Float16 a = (_Float16)v;
      qh[i] = a;
      ql[i] = (_Float16)((v - (float)a) * 2048.0f);
    } else {
      qh[i] = att_cvt<PREC>(v);
    }
  }
  f32x16 o_hi, o_lo;
#pragma unroll
  for (int r = 0; r < 16; ++r) o_hi[r] = o_lo[r] = 0.f;
  float m = -1e30f;
  const int vrow = min(l31, 8);  // A operand of O = V P: row = head dim (8 = the ones row, beyond: its copy, ignored)
  const _Float16 one16 = SPLIT ? (_Float16)1.0f : att_cvt<PREC>(1.0f), zero16 = __builtin_bit_cast(_Float16, (unsigned short)0);

  for (int j0 = 0; j0 < l; j0 += ATM_KT) {
    const int kt = min(ATM_KT, l - j0);
    __syncthreads();
    if constexpr (BLK) {  // one key per thread: its k and v vectors are one piece each
      for (int j = tid; j < ATM_KT; j += 64 * ATM_NW) {
        float kv[8], vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kv[i] = vv[i] = 0.f;
        if (j < kt) {
          load8(kb + (size_t)(j0 + j) * 8 * ESZ, kv);
          load8(vb + (size_t)(j0 + j) * 8 * ESZ, vv);
        }
        att_half8 k8, k8l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if constexpr (SPLIT) {
            const _Float16 ka = (_Float16)kv[i], va = (_Float16)vv[i];
            k8[i] = ka;
            k8l[i] = (_Float16)((kv[i] - (float)ka) * 2048.0f);
            Vh[i * ATM_VSTR + j] = va;
            Vl[i * ATM_VSTR + j] = (_Float16)((vv[i] - (float)va) * 2048.0f);
          } else {
            k8[i] = att_cvt<PREC>(kv[i]);
            Vh[i * ATM_VSTR + j] = att_cvt<PREC>(vv[i]);
          }
        }
        *reinterpret_cast<att_half8*>(&Kh[j * 8]) = k8;
        if constexpr (SPLIT) *reinterpret_cast<att_half8*>(&Kl[j * 8]) = k8l;
      }
    }
    for (int e = tid; e < (BLK ? 0 : 8 * ATM_KT); e += 64 * ATM_NW) {
      const int i = e / ATM_KT, j = e - i * ATM_KT;  // coalesced along the keys
      float kv = 0.f, vv = 0.f;
      if (j < kt) {
        kv = kp[(size_t)i * l + j0 + j];
        vv = vp[(size_t)i * l + j0 + j];
      }
      if constexpr (SPLIT) {
        const _Float16 ka = (_Float16)kv, va = (_Float16)vv;
        Kh[j * 8 + i] = ka;
        Kl[j * 8 + i] = (_Float16)((kv - (float)ka) * 2048.0f);
        Vh[i * ATM_VSTR + j] = va;
        Vl[i * ATM_VSTR + j] = (_Float16)((vv - (float)va) * 2048.0f);
      } else {
        Kh[j * 8 + i] = att_cvt<PREC>(kv);
        Vh[i * ATM_VSTR + j] = att_cvt<PREC>(vv);
      }
    }
    for (int j = tid; j < ATM_KT; j += 64 * ATM_NW) {
      Vh[8 * ATM_VSTR + j] = j < kt ? one16 : zero16;
      if constexpr (SPLIT) Vl[8 * ATM_VSTR + j] = (_Float16)0.0f;
    }
    __syncthreads();
    if (!active) continue;
    // S^T tile = 32 keys x 32 queries; the NEXT tile's MFMAs are issued before this tile's softmax arithmetic
    // so that the matrix pipe works under it
    auto s_tile = [&](int t, f32x16& s_hi, f32x16& s_lo) {
      const att_half4 kh = *reinterpret_cast<const att_half4*>(&Kh[(t + l31) * 8 + 4 * half]);
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if constexpr (SPLIT) {
        const att_half4 kl = *reinterpret_cast<const att_half4*>(&Kl[(t + l31) * 8 + 4 * half]);
        s_hi = __builtin_amdgcn_mfma_f32_32x32x8f16(kh, qh, zero, 0, 0, 0);  // (C = the inline constant 0)
        s_lo = __builtin_amdgcn_mfma_f32_32x32x8f16(kh, ql, zero, 0, 0, 0);
        s_lo = __builtin_amdgcn_mfma_f32_32x32x8f16(kl, qh, s_lo, 0, 0, 0);
      } else {
        s_hi = att_mma8<PREC>(kh, qh, zero);
      }
    };
    // one tile's softmax and O += V P from the scores in (s_hi, s_lo)
    auto pv_tile = [&](int t, const f32x16& s_hi, const f32x16& s_lo) {
      float sv[16];
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (SPLIT) sv[r] = s_hi[r] + s_lo[r] * (1.0f / 2048.0f);
        else sv[r] = s_hi[r];
        mx = fmaxf(mx, sv[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // the query's other 16 keys live in lane ^ 32
      const float mn = fmaxf(m, mx);
      const float sc = __builtin_amdgcn_exp2f(m - mn);  // (bare v_exp_f32: arguments <= 0, a flushed denormal is 0)
      m = mn;
#pragma unroll
      for (int r = 0; r < 5; ++r) {  // rows 0..3 (+4 half) = the head dims, row 8 (r = 4, half 0) = the denominator
        o_hi[r] *= sc;
        if constexpr (SPLIT) o_lo[r] *= sc;
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        // Registers 8 b .. 8 b + 7 are keys {0..3, 8..11} + 4 half of the tile's b-th 16 keys: as the B operand of a
        // 16-deep MFMA step they only need V (the A operand) read in the same key order.
        att_half8 ph, pl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float pv = __builtin_amdgcn_exp2f(sv[8 * b + i] - mn);
          if constexpr (SPLIT) {
            const _Float16 a = (_Float16)pv;
            ph[i] = a;
            pl[i] = (_Float16)((pv - (float)a) * 2048.0f);
          } else {
            ph[i] = att_cvt<PREC>(pv);
          }
        }
        const _Float16* vhp = &Vh[vrow * ATM_VSTR + t + 16 * b + 4 * half];
        const att_half4 vh0 = *reinterpret_cast<const att_half4*>(vhp), vh1 = *reinterpret_cast<const att_half4*>(vhp + 8);
        const att_half8 vh = {vh0[0], vh0[1], vh0[2], vh0[3], vh1[0], vh1[1], vh1[2], vh1[3]};
        if constexpr (SPLIT) {
          const _Float16* vlp = &Vl[vrow * ATM_VSTR + t + 16 * b + 4 * half];
          const att_half4 vl0 = *reinterpret_cast<const att_half4*>(vlp), vl1 = *reinterpret_cast<const att_half4*>(vlp + 8);
          const att_half8 vl = {vl0[0], vl0[1], vl0[2], vl0[3], vl1[0], vl1[1], vl1[2], vl1[3]};
          o_hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o_hi, 0, 0, 0);
          o_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o_lo, 0, 0, 0);
          o_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o_lo, 0, 0, 0);
        } else {
          o_hi = mma16<(PREC == 0 ? 2 : PREC)>(vh, ph, o_hi);
        }
      }
    };
    // two score sets alternate (no register copies): while one tile's arithmetic runs, the other's MFMAs are in flight
    f32x16 a_hi, a_lo, b_hi, b_lo;
    s_tile(0, a_hi, a_lo);
    int t = 0;
    for (; t + 64 <= kt; t += 64) {  // (kt % 32 == 0)
      s_tile(t + 32, b_hi, b_lo);
      pv_tile(t, a_hi, a_lo);
      s_tile(min(t + 64, kt - 32), a_hi, a_lo);  // (past the end: a tile that is not used)
      pv_tile(t + 32, b_hi, b_lo);
    }
    if (t < kt) pv_tile(t, a_hi, a_lo);  // odd tile count
  }
  if (!active) return;
  // row 8 of the tile (register 4 of the lower half-wave) is the denominator; the upper half-wave fetches it
  const float den_lo = SPLIT ? o_hi[4] + o_lo[4] * (1.0f / 2048.0f) : o_hi[4];  // (meaningful in the lower half-wave only)
  const float den_x = __shfl_xor(den_lo, 32, 64);
  const float den = half ? den_x : den_lo;
  const float inv = 1.0f / den;
  float* op = out + ((size_t)n * c + h * 8) * l;
  if (q0 + l31 < l) {
    if constexpr (BLK) {  // this lane's four head dims of its query: one 16- / 8-byte piece of the token's channel block
      float o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (SPLIT ? o_hi[r] + o_lo[r] * (1.0f / 2048.0f) : o_hi[r]) * inv;
      char* ob = reinterpret_cast<char*>(out) + ((((size_t)n * c + h * 8) * l + (size_t)(q0 + l31) * 8) + 4 * half) * ESZ;
      if constexpr (E16) *reinterpret_cast<att_u2*>(ob) = att_u2{pack2<PREC>(o4[0], o4[1]), pack2<PREC>(o4[2], o4[3])};
      else *reinterpret_cast<float4*>(ob) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
#pragma unroll
    for (int r = 0; r < (BLK ? 0 : 4); ++r)
      op[(size_t)(r + 4 * half) * l + q0 + l31] = (SPLIT ? o_hi[r] + o_lo[r] * (1.0f / 2048.0f) : o_hi[r]) * inv;
    if (lse && half == 0) lse[((size_t)n * heads + h) * l + q0 + l31] = m + log2f(den);
  }
}

static int g_att_mfma = 1;  // head_dim 8 on the matrix cores (tuning key 14: A/B against the VALU kernel)
void attention_set_mfma(int v) { g_att_mfma = v; }
static int g_att_bwd_split = 1;  // the fp32 tape's attention backward on the matrix cores, fp16x2 split (tuning key 38: A/B against the VALU kernels)
void attention_set_bwd_split(int v) { g_att_bwd_split = v; }
static int g_att_blocked = 1;  // the plan keeps q, k, v and the attention output channel-blocked (tuning key 25)
void attention_set_blocked(int v) { g_att_blocked = v; }
bool attention_blocked_ok(int c, int heads, int l) {
  return g_att_mfma && g_att_blocked && heads > 0 && c % heads == 0 && c / heads == 8 && l % 32 == 0;
}

// exact: keep off the fp16x2-split matrix-core kernel (q, k, v beyond fp16's range: the plan's range guard)
// dt: dsg_dtype of the products (DSG_F32 = the fp32-class split)
template <int D>
static int launch_attention(const float* qkv, float* out, float* lse, int n, int c, int heads, int l, hipStream_t st,
                            bool exact, int dt = DSG_F32) {
  // scores are kept in the log2 domain: q is pre-scaled by log2(e)/sqrt(D)
  const float qscale = 1.4426950408889634f / sqrtf((float)D);
  if (D == 8 && g_att_mfma && !exact && l % 32 == 0) {
    const dim3 grid(cdiv(l, 32 * ATM_NW) * heads * n), block(64 * ATM_NW);
    if (dt == DSG_BF16) hipLaunchKernelGGL(attention_mfma8_kernel<1>, grid, block, 0, st, qkv, out, lse, c, heads, l, qscale);
    else if (dt == DSG_F16) hipLaunchKernelGGL(attention_mfma8_kernel<2>, grid, block, 0, st, qkv, out, lse, c, heads, l, qscale);
    else hipLaunchKernelGGL(attention_mfma8_kernel<0>, grid, block, 0, st, qkv, out, lse, c, heads, l, qscale);
    DSG_LAUNCH_CHECK();
    return DSG_OK;
  }
  const long work = (long)n * heads * l;
  int qpt = 4;
  if (work / (256 * 4) < 512) qpt = 2;
  if (work / (256 * 2) < 512) qpt = 1;
  if (D > 16) qpt = 1;
  dim3 grid(cdiv(l, 256 * qpt), heads, n);
  if constexpr (D <= 16) {  // wider heads keep one query per thread (register budget)
    if (qpt == 4) {
      hipLaunchKernelGGL((attention_kernel<D, 4>), grid, dim3(256), 0, st, qkv, out, lse, c, heads, l, qscale);
      DSG_LAUNCH_CHECK();
      return DSG_OK;
    }
    if (qpt == 2) {
      hipLaunchKernelGGL((attention_kernel<D, 2>), grid, dim3(256), 0, st, qkv, out, lse, c, heads, l, qscale);
      DSG_LAUNCH_CHECK();
      return DSG_OK;
    }
  }
  hipLaunchKernelGGL((attention_kernel<D, 1>), grid, dim3(256), 0, st, qkv, out, lse, c, heads, l, qscale);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

}  // namespace dsg

static int attention_fwd_impl(const float* qkv, float* out, float* lse, int32_t n, int32_t c, int32_t heads, int32_t l,
                              void* stream, bool exact = false, int dt = DSG_F32) {
  DSG_CHECK_ARG(qkv && out, "dsg_attention_fwd: NULL pointer");
  DSG_CHECK_ARG(n > 0 && c > 0 && heads > 0 && l > 0, "dsg_attention_fwd: bad dims");
  DSG_CHECK_ARG(c % heads == 0, "dsg_attention_fwd: channels (%d) not divisible by heads (%d)", c, heads);
  DSG_CHECK_ARG(heads <= 65535 && n <= 65535, "dsg_attention_fwd: grid too large");
  const int d = c / heads;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (d) {
    case 8: return dsg::launch_attention<8>(qkv, out, lse, n, c, heads, l, st, exact, dt);
    case 16: return dsg::launch_attention<16>(qkv, out, lse, n, c, heads, l, st, exact);
    case 32: return dsg::launch_attention<32>(qkv, out, lse, n, c, heads, l, st, exact);
    case 64: return dsg::launch_attention<64>(qkv, out, lse, n, c, heads, l, st, exact);
    default:
      return dsg::fail(DSG_ERR_UNSUPPORTED_SHAPE, "dsg_attention_fwd: head_dim %d not in {8,16,32,64}", d);
  }
}

namespace dsg {
// the plan's range guard: q / k / v projections whose weights left the split's range run the fp32 VALU kernel
int attention_fwd_exact(const float* qkv, float* out, int n, int c, int heads, int l, hipStream_t st) {
  return attention_fwd_impl(qkv, out, nullptr, n, c, heads, l, st, true);
}
}  // namespace dsg

DSG_API int dsg_attention_fwd(const float* qkv, float* out, int32_t n, int32_t c, int32_t heads, int32_t l,
                              void* stream) {
  return attention_fwd_impl(qkv, out, nullptr, n, c, heads, l, stream);
}

DSG_API int dsg_attention_fwd_dt(const float* qkv, float* out, int32_t n, int32_t c, int32_t heads, int32_t l,
                                 int32_t dtype, void* stream) {
  DSG_CHECK_ARG(dtype >= DSG_F32 && dtype <= DSG_F16, "dsg_attention_fwd_dt: bad dtype %d", dtype);
  return attention_fwd_impl(qkv, out, nullptr, n, c, heads, l, stream, false, dtype);
}

// the same on channel-blocked tensors (see attention_mfma8_kernel<PREC, BLK>): head_dim 8, l % 32 == 0 only
DSG_API int dsg_attention_fwd_blocked(const void* qkv, void* out, int32_t n, int32_t c, int32_t heads, int32_t l,
                                      int32_t dtype, void* stream) {
  DSG_CHECK_ARG(qkv && out, "dsg_attention_fwd_blocked: NULL pointer");
  DSG_CHECK_ARG(dtype >= DSG_F32 && dtype <= DSG_F16, "dsg_attention_fwd_blocked: bad dtype %d", dtype);
  DSG_CHECK_ARG(n > 0 && c > 0 && heads > 0 && l > 0 && c % heads == 0, "dsg_attention_fwd_blocked: bad dims");
  DSG_CHECK_SHAPE(c / heads == 8 && l % 32 == 0,
                  "dsg_attention_fwd_blocked: head_dim %d / %d tokens (needs head_dim 8 = one channel block, tokens %% 32 == 0)",
                  c / heads, l);
  const float qscale = 1.4426950408889634f / sqrtf(8.0f);
  const dim3 grid(dsg::cdiv(l, 32 * dsg::ATM_NW) * heads * n), block(64 * dsg::ATM_NW);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float* q = static_cast<const float*>(qkv);
  float* o = static_cast<float*>(out);
  if (dtype == DSG_BF16) hipLaunchKernelGGL((dsg::attention_mfma8_kernel<1, true>), grid, block, 0, st, q, o, nullptr, c, heads, l, qscale);
  else if (dtype == DSG_F16) hipLaunchKernelGGL((dsg::attention_mfma8_kernel<2, true>), grid, block, 0, st, q, o, nullptr, c, heads, l, qscale);
  else hipLaunchKernelGGL((dsg::attention_mfma8_kernel<0, true>), grid, block, 0, st, q, o, nullptr, c, heads, l, qscale);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

DSG_API int dsg_attention_fwd_train(const float* qkv, float* out, float* lse, int32_t n, int32_t c, int32_t heads,
                                    int32_t l, void* stream) {
  DSG_CHECK_ARG(lse != nullptr, "dsg_attention_fwd_train: lse is NULL");
  return attention_fwd_impl(qkv, out, lse, n, c, heads, l, stream);
}

// the same in the arithmetic of the mixed-precision tape (DSG_BF16: the forward of dsg_attention_bwd_dt's matrix-core kernels --
// both round q * scale and k alike, so the backward recomputes exactly the probabilities the forward used)
DSG_API int dsg_attention_fwd_train_dt(const float* qkv, float* out, float* lse, int32_t n, int32_t c, int32_t heads,
                                       int32_t l, int32_t dtype, void* stream) {
  DSG_CHECK_ARG(lse != nullptr, "dsg_attention_fwd_train_dt: lse is NULL");
  DSG_CHECK_ARG(dtype >= DSG_F32 && dtype <= DSG_F16, "dsg_attention_fwd_train_dt: bad dtype %d", dtype);
  return attention_fwd_impl(qkv, out, lse, n, c, heads, l, stream, false, dtype);
}

// ---------------------------------------------------------------------------------------------------
// Backward of the attention core (flash-style recompute from the saved log-sum-exp):
//   P = exp2(s - lse), dP = dO . V, dS = P * (dP - D), D = rowsum(dO * O)
//   dQ = dS K * scale, dK = dS^T Q * scale, dV = P^T dO
// Kernel A: one thread per query row (keys/values streamed through LDS) -> dQ and D.
// Kernel B: one thread per key row (queries / dO / lse / D streamed through LDS) -> dK, dV.
// ---------------------------------------------------------------------------------------------------
namespace dsg {

typedef float att_f2 __attribute__((ext_vector_type(2)));

template <int D>
__global__ __launch_bounds__(256) void attention_bwd_dq_kernel(const float* __restrict__ qkv,
                                                               const float* __restrict__ o,
                                                               const float* __restrict__ dout,
                                                               const float* __restrict__ lse, float* __restrict__ dqkv,
                                                               float* __restrict__ dsum, int c, int heads, int l,
                                                               float qscale) {
  constexpr int KT = ATT_LDS_FLOATS / D;
  __shared__ __attribute__((aligned(16))) float KVl[2 * KT * D];
  float* Kl = KVl;
  float* Vl = KVl + KT * D;
  const int tid = threadIdx.x;
  const int h = blockIdx.y, n = blockIdx.z;
  const float* qp = qkv + ((size_t)n * 3 * c + h * D) * l;
  const float* kp = qp + (size_t)c * l;
  const float* vp = kp + (size_t)c * l;
  const size_t obase = ((size_t)n * c + h * D) * l;
  const int qi = blockIdx.x * 256 + tid;
  const int qc = min(qi, l - 1);
  float q[D], dO[D], dq[D];
  att_f2 dq2[D];  // (even keys, odd keys)
#pragma unroll
  for (int i = 0; i < D; ++i) dq2[i] = att_f2{0.f, 0.f};
  float dd = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    q[i] = qp[(size_t)i * l + qc] * qscale;
    dO[i] = dout[obase + (size_t)i * l + qc];
    dd = fmaf(dO[i], o[obase + (size_t)i * l + qc], dd);
    dq[i] = 0.f;
  }
  const float ls = lse[((size_t)n * heads + h) * l + qc];
  for (int j0 = 0; j0 < l; j0 += KT) {
    const int kt = min(KT, l - j0);
    __syncthreads();
    for (int e = tid; e < KT * D; e += 256) {
      const int i = e / KT, j = e - i * KT;
      float kv = 0.f, vv = 0.f;
      if (j < kt) {
        kv = kp[(size_t)i * l + j0 + j];
        vv = vp[(size_t)i * l + j0 + j];
      }
      // key PAIRS side by side: [pair][dim][2] -- the loop below works on two keys per packed fp32 instruction
      Kl[((j >> 1) * D + i) * 2 + (j & 1)] = kv;
      Vl[((j >> 1) * D + i) * 2 + (j & 1)] = vv;
    }
    __syncthreads();
    // (a padded key has K = V = 0: its probability is not zero, but everything it is multiplied into dq by is)
    // (the pair's K and V rows are fetched one iteration ahead into a second register set -- consumed as they are read,
    // the broadcast LDS reads' latency was most of an iteration; two sets taking turns, no copies)
    const int npair = (kt + 1) / 2;
    const att_f2* K2 = reinterpret_cast<const att_f2*>(Kl);
    const att_f2* V2 = reinterpret_cast<const att_f2*>(Vl);
    auto fetch = [&](int jp, att_f2 (&kr)[D], att_f2 (&vr)[D]) {
      const int jc = min(jp, npair - 1);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        kr[i] = K2[jc * D + i];
        vr[i] = V2[jc * D + i];
      }
    };
    auto pair = [&](const att_f2 (&kr)[D], const att_f2 (&vr)[D]) {
      att_f2 s = {0.f, 0.f}, dp = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < D; ++i) {
        s = __builtin_elementwise_fma(att_f2{q[i], q[i]}, kr[i], s);
        dp = __builtin_elementwise_fma(att_f2{dO[i], dO[i]}, vr[i], dp);
      }
      const att_f2 pr = {__builtin_amdgcn_exp2f(s.x - ls), __builtin_amdgcn_exp2f(s.y - ls)};  // (arguments <= ~0)
      const att_f2 ds = pr * (dp - att_f2{dd, dd});
#pragma unroll
      for (int i = 0; i < D; ++i) dq2[i] = __builtin_elementwise_fma(ds, kr[i], dq2[i]);
    };
    att_f2 ka[D], va[D], kb[D], vb[D];
    fetch(0, ka, va);
    int jp = 0;
    for (; jp + 1 < npair; jp += 2) {
      fetch(jp + 1, kb, vb);
      pair(ka, va);
      fetch(jp + 2, ka, va);
      pair(kb, vb);
    }
    if (jp < npair) pair(ka, va);
  }
#pragma unroll
  for (int i = 0; i < D; ++i) dq[i] = dq2[i].x + dq2[i].y;
  if (qi < l) {
    // qscale = log2(e)/sqrt(D); the softmax scale alone is 1/sqrt(D)
    const float sm = qscale * 0.6931471805599453f;
    float* dqp = dqkv + ((size_t)n * 3 * c + h * D) * l;
#pragma unroll
    for (int i = 0; i < D; ++i) dqp[(size_t)i * l + qi] = dq[i] * sm;
    dsum[((size_t)n * heads + h) * l + qi] = dd;
  }
}

template <int D>
__global__ __launch_bounds__(256) void attention_bwd_dkv_kernel(const float* __restrict__ qkv,
                                                                const float* __restrict__ dout,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ dsum,
                                                                float* __restrict__ dqkv, int c, int heads, int l,
                                                                float qscale) {
  constexpr int QT = (ATT_LDS_FLOATS / (D + 1)) & ~1;  // queries per LDS tile (even: they sit in pairs): q[D], dO[D], lse, D-sum
  __shared__ __attribute__((aligned(16))) float Sm[2 * QT * (D + 1)];
  float* Ql = Sm;                  // [QT][D+1]: q (prescaled) then lse
  float* Gl = Sm + QT * (D + 1);   // [QT][D+1]: dO then dsum
  const int tid = threadIdx.x;
  const int h = blockIdx.y, n = blockIdx.z;
  const float* qp = qkv + ((size_t)n * 3 * c + h * D) * l;
  const float* kp = qp + (size_t)c * l;
  const float* vp = kp + (size_t)c * l;
  const size_t obase = ((size_t)n * c + h * D) * l;
  const size_t lbase = ((size_t)n * heads + h) * l;
  const int ki = blockIdx.x * 256 + tid;
  const int kc = min(ki, l - 1);
  float k[D], v[D], dk[D], dv[D];
  att_f2 dk2[D], dv2[D];  // (even queries, odd queries)
#pragma unroll
  for (int i = 0; i < D; ++i) dk2[i] = dv2[i] = att_f2{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < D; ++i) {
    k[i] = kp[(size_t)i * l + kc];
    v[i] = vp[(size_t)i * l + kc];
    dk[i] = 0.f;
    dv[i] = 0.f;
  }
  for (int j0 = 0; j0 < l; j0 += QT) {
    const int qt = min(QT, l - j0);
    __syncthreads();
    for (int e = tid; e < QT * (D + 1); e += 256) {
      const int i = e / QT, j = e - i * QT;
      float a = 0.f, b = 0.f;
      if (j < qt) {
        if (i < D) {
          a = qp[(size_t)i * l + j0 + j] * qscale;
          b = dout[obase + (size_t)i * l + j0 + j];
        } else {
          a = lse[lbase + j0 + j];
          b = dsum[lbase + j0 + j];
        }
      }
      // query PAIRS side by side: [pair][dim + 1][2]
      Ql[((j >> 1) * (D + 1) + i) * 2 + (j & 1)] = a;
      Gl[((j >> 1) * (D + 1) + i) * 2 + (j & 1)] = b;
    }
    __syncthreads();
    // (a padded query has q = dO = lse = D-sum = 0: whatever its probability, it adds 0 to dk and dv)
    const int npair = (qt + 1) / 2;
    const att_f2* Q2 = reinterpret_cast<const att_f2*>(Ql);
    const att_f2* G2 = reinterpret_cast<const att_f2*>(Gl);
    auto fetch = [&](int jp, att_f2 (&qr)[D + 1], att_f2 (&gr)[D + 1]) {
      const int jc = min(jp, npair - 1);
#pragma unroll
      for (int i = 0; i <= D; ++i) {
        qr[i] = Q2[jc * (D + 1) + i];
        gr[i] = G2[jc * (D + 1) + i];
      }
    };
    auto pair = [&](const att_f2 (&qr)[D + 1], const att_f2 (&gr)[D + 1]) {
      att_f2 s = {0.f, 0.f}, dp = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < D; ++i) {
        s = __builtin_elementwise_fma(qr[i], att_f2{k[i], k[i]}, s);
        dp = __builtin_elementwise_fma(gr[i], att_f2{v[i], v[i]}, dp);
      }
      const att_f2 e = s - qr[D];
      const att_f2 pr = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
      const att_f2 ds = pr * (dp - gr[D]);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        dv2[i] = __builtin_elementwise_fma(pr, gr[i], dv2[i]);
        dk2[i] = __builtin_elementwise_fma(ds, qr[i], dk2[i]);  // qr is prescaled by log2(e)/sqrt(D): undone below
      }
    };
    att_f2 qa[D + 1], ga[D + 1], qb[D + 1], gb[D + 1];
    fetch(0, qa, ga);
    int jp = 0;
    for (; jp + 1 < npair; jp += 2) {
      fetch(jp + 1, qb, gb);
      pair(qa, ga);
      fetch(jp + 2, qa, ga);
      pair(qb, gb);
    }
    if (jp < npair) pair(qa, ga);
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    dk[i] = dk2[i].x + dk2[i].y;
    dv[i] = dv2[i].x + dv2[i].y;
  }
  if (ki < l) {
    float* dkp = dqkv + ((size_t)n * 3 * c + c + h * D) * l;
    float* dvp = dkp + (size_t)c * l;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      dkp[(size_t)i * l + ki] = dk[i] * 0.6931471805599453f;
      dvp[(size_t)i * l + ki] = dv[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The same backward on the matrix cores for head_dim 8 in the mixed-precision tapes: q, k, v, dO, P and dS are rounded
// once to 16 bits, one MFMA per product (PREC 1: bf16.  PREC 2, the fp16 tape: dS = P (dP - D) is a probability times a
// gradient, 1e-3 x 1e-6 without a loss scale -- below fp16's normal range, flushed bits that the test of a whole training step
// caught.  The gradients are linear in dO, so dO is multiplied by a power of two that brings its largest entry to [1, 2) before
// it is rounded and the results are divided by it: per query in kernel A, per (image, head) in kernel B, whose sums run over the
// queries), fp32 scores / accumulators -- torch.autocast's split for the attention core (the fp32 tape
// keeps the exact VALU kernels above: with fp16-pair operands the split arithmetic would cost what the packed-fp32 loops do).
// Layouts are the forward kernel's (attention_mfma8_kernel): a 32 x 32 tile of S^T = K^T Q leaves a lane with ONE column and 16
// rows, and registers 8b .. 8b + 7 are the B operand of a 16-deep step over those rows.
//   kernel A (dQ): columns = this wave's 32 queries (q, dO, lse, D = rowsum(dO * O) live in registers), rows = keys from LDS:
//                  S^T = K^T Q, dP^T = V^T dO, dS = P (dP - D), dQ^T += K^T(as [d][key]) dS
//   kernel B (dK, dV): columns = this wave's 32 keys, rows = queries from LDS (with their lse and D):
//                  S = Q^T K, dP = dO^T V, dV^T += dO(as [d][query]) P, dK^T += Q(as [d][query]) dS
// ---------------------------------------------------------------------------------------------------
// (value, (value - hi) * 2^11) as an fp16 pair: the fp16x2 split of the PREC 0 kernels (x == hi + lo * 2^-11 to 2^-22 relative)
// (a macro: the targets are vector ELEMENTS, which a reference parameter cannot bind to)
#define ATT_SPLIT(v, hi, lo)                               \
  do {                                                     \
    const float sv_ = (v);                                 \
    const _Float16 sh_ = (_Float16)sv_;                    \
    (hi) = sh_;                                            \
    (lo) = (_Float16)((sv_ - (float)sh_) * 2048.0f);       \
  } while (0)

// PREC 0 (round 6; the fp32 tape's attention backward, training_pipeline.py:86 through the mid block's Attention): every product
// as the fp16x2 split -- a.b ~ a1.b1 + 2^-11 (a1.b2 + a2.b1), three MFMAs, the cross terms on accumulators of their own -- with
// dO brought to [1, 2) by a power of two first (the pieces are fp16: unscaled gradients of 1e-6 would land in its subnormals).
// fp32-class accuracy (tests/test_gpu_train_ops.py: 2e-5 against torch autograd, the VALU kernels' own tolerance) on the
// matrix cores: the fp32 tape's two VALU kernels were 6.5 ms of the configs[2] step.
template <int PREC>
__global__ __launch_bounds__(64 * ATM_NW, 2) void attention_bwd_dq_mfma8_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                                               const float* __restrict__ dout, const float* __restrict__ lse,
                                                                               float* __restrict__ dqkv, float* __restrict__ dsum, int c,
                                                                               int heads, int l, float qscale) {
  constexpr bool SPLIT = PREC == 0;
  constexpr bool SCALE = PREC != 1;  // fp16 pieces: dO is brought to [1, 2) by a power of two before it is rounded (see above)
  __shared__ __attribute__((aligned(16))) _Float16 Kh[ATM_KT * 8], Vk[ATM_KT * 8];   // [key][d]: A operands of S^T and dP^T
  __shared__ __attribute__((aligned(16))) _Float16 Kd[8 * ATM_VSTR];                  // [d][key]: A operand of dQ^T += K dS
  __shared__ __attribute__((aligned(16))) _Float16 Kl[SPLIT ? ATM_KT * 8 : 8], Vkl[SPLIT ? ATM_KT * 8 : 8], Kdl[SPLIT ? 8 * ATM_VSTR : 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int qtiles = (l + 32 * ATM_NW - 1) / (32 * ATM_NW);
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // (a head's tiles behind one L2)
  const int qt = bid % qtiles, hn = bid / qtiles;
  const int h = hn % heads, n = hn / heads;
  const float* qp = qkv + ((size_t)n * 3 * c + h * 8) * l;
  const float* kp = qp + (size_t)c * l;
  const float* vp = kp + (size_t)c * l;
  const size_t obase = ((size_t)n * c + h * 8) * l;
  const int q0 = (qt * ATM_NW + wave) * 32;
  const bool active = q0 < l;
  const int qi = min(q0 + l31, l - 1);
  att_half4 qh, dh, ql, dl;
  float dpart = 0.f;
  float dov4[4], amax = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const size_t at = (size_t)(4 * half + i) * l + qi;
    dov4[i] = dout[obase + at];
    if constexpr (SPLIT) ATT_SPLIT(qp[at] * qscale, qh[i], ql[i]);
    else qh[i] = att_cvt<PREC>(qp[at] * qscale);
    dpart = fmaf(dov4[i], o[obase + at], dpart);
    amax = fmaxf(amax, fabsf(dov4[i]));
  }
  // this query's power-of-two scale (lanes l31 and l31 + 32 hold its two halves): dQ_i is linear in dO_i, undone at the end
  float qs = 1.f, qs_inv = 1.f;
  if constexpr (SCALE) {
    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
    if (amax > 0.f && amax < 3.0e38f) {
      int ex;
      (void)frexpf(amax, &ex);
      qs = ldexpf(1.0f, 1 - ex);
      qs_inv = ldexpf(1.0f, ex - 1);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (SPLIT) ATT_SPLIT(dov4[i] * qs, dh[i], dl[i]);
    else dh[i] = att_cvt<PREC>(dov4[i] * qs);
  }
  const float dd_true = dpart + __shfl_xor(dpart, 32, 64);   // D = rowsum(dO * O) of this lane's query
  const float dd = dd_true * qs;
  const float ls = lse[((size_t)n * heads + h) * l + qi];
  f32x16 dq, dq_lo;
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[r] = dq_lo[r] = 0.f;
  const int vrow = min(l31, 7);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < l; j0 += ATM_KT) {
    const int kt = min(ATM_KT, l - j0);
    __syncthreads();
    for (int e = tid; e < 8 * ATM_KT; e += 64 * ATM_NW) {
      const int i = e / ATM_KT, j = e - i * ATM_KT;  // coalesced along the keys
      float kv = 0.f, vv = 0.f;
      if (j < kt) {
        kv = kp[(size_t)i * l + j0 + j];
        vv = vp[(size_t)i * l + j0 + j];
      }
      if constexpr (SPLIT) {
        _Float16 kh16, kl16, vh16, vl16;
        ATT_SPLIT(kv, kh16, kl16);
        ATT_SPLIT(vv, vh16, vl16);
        Kh[j * 8 + i] = kh16; Kl[j * 8 + i] = kl16;
        Kd[i * ATM_VSTR + j] = kh16; Kdl[i * ATM_VSTR + j] = kl16;
        Vk[j * 8 + i] = vh16; Vkl[j * 8 + i] = vl16;
      } else {
        Kh[j * 8 + i] = att_cvt<PREC>(kv);
        Kd[i * ATM_VSTR + j] = att_cvt<PREC>(kv);
        Vk[j * 8 + i] = att_cvt<PREC>(vv);
      }
    }
    __syncthreads();
    if (!active) continue;
    for (int t = 0; t < kt; t += 32) {
      const att_half4 ka = *reinterpret_cast<const att_half4*>(&Kh[(t + l31) * 8 + 4 * half]);
      const att_half4 va = *reinterpret_cast<const att_half4*>(&Vk[(t + l31) * 8 + 4 * half]);
      f32x16 sc = att_mma8<PREC>(ka, qh, zero);   // S^T: rows = keys, this lane's column = its query
      f32x16 dp = att_mma8<PREC>(va, dh, zero);   // dP^T
      if constexpr (SPLIT) {
        const att_half4 kal = *reinterpret_cast<const att_half4*>(&Kl[(t + l31) * 8 + 4 * half]);
        const att_half4 val = *reinterpret_cast<const att_half4*>(&Vkl[(t + l31) * 8 + 4 * half]);
        f32x16 s2 = att_mma8<PREC>(ka, ql, zero), p2 = att_mma8<PREC>(va, dl, zero);
        s2 = att_mma8<PREC>(kal, qh, s2);
        p2 = att_mma8<PREC>(val, dh, p2);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sc[r] += s2[r] * (1.0f / 2048.0f);
          dp[r] += p2[r] * (1.0f / 2048.0f);
        }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        att_half8 dsh, dsl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float pv = __builtin_amdgcn_exp2f(sc[8 * b + i] - ls);
          if constexpr (SPLIT) ATT_SPLIT(pv * (dp[8 * b + i] - dd), dsh[i], dsl[i]);
          else dsh[i] = att_cvt<PREC>(pv * (dp[8 * b + i] - dd));
        }
        const _Float16* kdp = &Kd[vrow * ATM_VSTR + t + 16 * b + 4 * half];
        const att_half4 k0 = *reinterpret_cast<const att_half4*>(kdp), k1 = *reinterpret_cast<const att_half4*>(kdp + 8);
        const att_half8 kd = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
        dq = mma16<PREC>(kd, dsh, dq);
        if constexpr (SPLIT) {
          const _Float16* klp = &Kdl[vrow * ATM_VSTR + t + 16 * b + 4 * half];
          const att_half4 m0 = *reinterpret_cast<const att_half4*>(klp), m1 = *reinterpret_cast<const att_half4*>(klp + 8);
          const att_half8 kdl = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
          dq_lo = mma16<PREC>(kd, dsl, dq_lo);
          dq_lo = mma16<PREC>(kdl, dsh, dq_lo);
        }
      }
    }
  }
  if (!active || q0 + l31 >= l) return;
  const float sm = qscale * 0.6931471805599453f * qs_inv;  // qscale = log2(e)/sqrt(D); the softmax scale alone is 1/sqrt(D)
  float* dqp = dqkv + ((size_t)n * 3 * c + h * 8) * l;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    dqp[(size_t)(r + 4 * half) * l + q0 + l31] = (SPLIT ? dq[r] + dq_lo[r] * (1.0f / 2048.0f) : dq[r]) * sm;
  if (half == 0) dsum[((size_t)n * heads + h) * l + q0 + l31] = dd_true;
}

template <int PREC>
__global__ __launch_bounds__(64 * ATM_NW, PREC == 0 ? 1 : 2) void attention_bwd_dkv_mfma8_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                                const float* __restrict__ lse, const float* __restrict__ dsum,
                                                                                float* __restrict__ dqkv, int c, int heads, int l,
                                                                                float qscale) {
  constexpr bool SPLIT = PREC == 0;  // fp32-class: every product as the fp16x2 split (see the dq kernel)
  constexpr bool SCALE = PREC != 1;  // fp16 pieces: dO is brought to [1, 2) by a power of two before it is rounded (see above)
  __shared__ __attribute__((aligned(16))) _Float16 Qh[ATM_KT * 8], Gh[ATM_KT * 8];            // [query][d]: A operands of S and dP
  __shared__ __attribute__((aligned(16))) _Float16 Qd[8 * ATM_VSTR], Gd[8 * ATM_VSTR];        // [d][query]: A operands of dK^T, dV^T
  __shared__ __attribute__((aligned(16))) _Float16 Ql[SPLIT ? ATM_KT * 8 : 8], Gl[SPLIT ? ATM_KT * 8 : 8];
  __shared__ __attribute__((aligned(16))) _Float16 Qdl[SPLIT ? 8 * ATM_VSTR : 8], Gdl[SPLIT ? 8 * ATM_VSTR : 8];
  __shared__ __attribute__((aligned(16))) float Ls[ATM_KT], Ds[ATM_KT];                        // lse and D of the tile's queries
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int ktiles = (l + 32 * ATM_NW - 1) / (32 * ATM_NW);
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int kt_i = bid % ktiles, hn = bid / ktiles;
  const int h = hn % heads, n = hn / heads;
  const float* qp = qkv + ((size_t)n * 3 * c + h * 8) * l;
  const float* kp = qp + (size_t)c * l;
  const float* vp = kp + (size_t)c * l;
  const size_t obase = ((size_t)n * c + h * 8) * l;
  const size_t lbase = ((size_t)n * heads + h) * l;
  const int k0 = (kt_i * ATM_NW + wave) * 32;
  const bool active = k0 < l;
  const int ki = min(k0 + l31, l - 1);
  att_half4 kh, vh, kl, vl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (SPLIT) {
      ATT_SPLIT(kp[(size_t)(4 * half + i) * l + ki], kh[i], kl[i]);
      ATT_SPLIT(vp[(size_t)(4 * half + i) * l + ki], vh[i], vl[i]);
    } else {
      kh[i] = att_cvt<PREC>(kp[(size_t)(4 * half + i) * l + ki]);
      vh[i] = att_cvt<PREC>(vp[(size_t)(4 * half + i) * l + ki]);
    }
  }
  // dK_j and dV_j sum over the queries: ONE scale for the head's whole dO (its 8 l values are contiguous; every workgroup of
  // the head finds the same maximum -- no atomics, no extra buffer); rows far below the maximum lose bits that do not show in the sum
  float gs = 1.f, gs_inv = 1.f;
  if constexpr (SCALE) {
    __shared__ float wmax[ATM_NW];
    float m = 0.f;
    for (int e = tid; e < 8 * l; e += 64 * ATM_NW) m = fmaxf(m, fabsf(dout[obase + e]));
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    m = wmax[0];
#pragma unroll
    for (int w = 1; w < ATM_NW; ++w) m = fmaxf(m, wmax[w]);
    if (m > 0.f && m < 3.0e38f) {
      int ex;
      (void)frexpf(m, &ex);
      gs = ldexpf(1.0f, 1 - ex);
      gs_inv = ldexpf(1.0f, ex - 1);
    }
  }
  f32x16 dk, dv, dk_lo, dv_lo;
#pragma unroll
  for (int r = 0; r < 16; ++r) dk[r] = dv[r] = dk_lo[r] = dv_lo[r] = 0.f;
  const int vrow = min(l31, 7);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < l; j0 += ATM_KT) {
    const int qt = min(ATM_KT, l - j0);
    __syncthreads();
    for (int e = tid; e < 8 * ATM_KT; e += 64 * ATM_NW) {
      const int i = e / ATM_KT, j = e - i * ATM_KT;  // coalesced along the queries
      float qv = 0.f, gv = 0.f;
      if (j < qt) {
        qv = qp[(size_t)i * l + j0 + j] * qscale;
        gv = dout[obase + (size_t)i * l + j0 + j] * gs;
      }
      if constexpr (SPLIT) {
        _Float16 q1, q2, g1, g2;
        ATT_SPLIT(qv, q1, q2);
        ATT_SPLIT(gv, g1, g2);
        Qh[j * 8 + i] = q1; Ql[j * 8 + i] = q2;
        Qd[i * ATM_VSTR + j] = q1; Qdl[i * ATM_VSTR + j] = q2;
        Gh[j * 8 + i] = g1; Gl[j * 8 + i] = g2;
        Gd[i * ATM_VSTR + j] = g1; Gdl[i * ATM_VSTR + j] = g2;
      } else {
        const _Float16 gb = att_cvt<PREC>(gv);
        Qh[j * 8 + i] = att_cvt<PREC>(qv);
        Qd[i * ATM_VSTR + j] = att_cvt<PREC>(qv);
        Gh[j * 8 + i] = gb;
        Gd[i * ATM_VSTR + j] = gb;
      }
    }
    for (int j = tid; j < ATM_KT; j += 64 * ATM_NW) {
      Ls[j] = j < qt ? lse[lbase + j0 + j] : 0.f;
      Ds[j] = j < qt ? dsum[lbase + j0 + j] * gs : 0.f;
    }
    __syncthreads();
    if (!active) continue;
    for (int t = 0; t < qt; t += 32) {
      const att_half4 qa = *reinterpret_cast<const att_half4*>(&Qh[(t + l31) * 8 + 4 * half]);
      const att_half4 ga = *reinterpret_cast<const att_half4*>(&Gh[(t + l31) * 8 + 4 * half]);
      f32x16 sc = att_mma8<PREC>(qa, kh, zero);   // S: rows = queries, this lane's column = its key
      f32x16 dp = att_mma8<PREC>(ga, vh, zero);   // dP
      if constexpr (SPLIT) {
        const att_half4 qal = *reinterpret_cast<const att_half4*>(&Ql[(t + l31) * 8 + 4 * half]);
        const att_half4 gal = *reinterpret_cast<const att_half4*>(&Gl[(t + l31) * 8 + 4 * half]);
        f32x16 s2 = att_mma8<PREC>(qa, kl, zero), p2 = att_mma8<PREC>(ga, vl, zero);
        s2 = att_mma8<PREC>(qal, kh, s2);
        p2 = att_mma8<PREC>(gal, vh, p2);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sc[r] += s2[r] * (1.0f / 2048.0f);
          dp[r] += p2[r] * (1.0f / 2048.0f);
        }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        // registers 8b .. 8b + 7 = queries t + 16 b + 4 half + {0..3, 8..11}
        const float4 l0 = *reinterpret_cast<const float4*>(&Ls[t + 16 * b + 4 * half]);
        const float4 l1 = *reinterpret_cast<const float4*>(&Ls[t + 16 * b + 4 * half + 8]);
        const float4 d0 = *reinterpret_cast<const float4*>(&Ds[t + 16 * b + 4 * half]);
        const float4 d1 = *reinterpret_cast<const float4*>(&Ds[t + 16 * b + 4 * half + 8]);
        const float lsr[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        const float ddr[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        att_half8 ph, dsh, pl, dsl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float pv = __builtin_amdgcn_exp2f(sc[8 * b + i] - lsr[i]);
          if constexpr (SPLIT) {
            ATT_SPLIT(pv, ph[i], pl[i]);
            ATT_SPLIT(pv * (dp[8 * b + i] - ddr[i]), dsh[i], dsl[i]);
          } else {
            ph[i] = att_cvt<PREC>(pv);
            dsh[i] = att_cvt<PREC>(pv * (dp[8 * b + i] - ddr[i]));
          }
        }
        const _Float16* gdp = &Gd[vrow * ATM_VSTR + t + 16 * b + 4 * half];
        const _Float16* qdp = &Qd[vrow * ATM_VSTR + t + 16 * b + 4 * half];
        const att_half4 g0 = *reinterpret_cast<const att_half4*>(gdp), g1 = *reinterpret_cast<const att_half4*>(gdp + 8);
        const att_half4 x0 = *reinterpret_cast<const att_half4*>(qdp), x1 = *reinterpret_cast<const att_half4*>(qdp + 8);
        const att_half8 gd = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const att_half8 qd = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        dv = mma16<PREC>(gd, ph, dv);
        dk = mma16<PREC>(qd, dsh, dk);
        if constexpr (SPLIT) {
          const _Float16* glp = &Gdl[vrow * ATM_VSTR + t + 16 * b + 4 * half];
          const _Float16* qlp = &Qdl[vrow * ATM_VSTR + t + 16 * b + 4 * half];
          const att_half4 h0 = *reinterpret_cast<const att_half4*>(glp), h1 = *reinterpret_cast<const att_half4*>(glp + 8);
          const att_half4 y0 = *reinterpret_cast<const att_half4*>(qlp), y1 = *reinterpret_cast<const att_half4*>(qlp + 8);
          const att_half8 gdl = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
          const att_half8 qdl = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
          dv_lo = mma16<PREC>(gd, pl, dv_lo);
          dv_lo = mma16<PREC>(gdl, ph, dv_lo);
          dk_lo = mma16<PREC>(qd, dsl, dk_lo);
          dk_lo = mma16<PREC>(qdl, dsh, dk_lo);
        }
      }
    }
  }
  if (!active || k0 + l31 >= l) return;
  float* dkp = dqkv + ((size_t)n * 3 * c + c + h * 8) * l;
  float* dvp = dkp + (size_t)c * l;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float dkr = SPLIT ? dk[r] + dk_lo[r] * (1.0f / 2048.0f) : dk[r], dvr = SPLIT ? dv[r] + dv_lo[r] * (1.0f / 2048.0f) : dv[r];
    dkp[(size_t)(r + 4 * half) * l + k0 + l31] = dkr * (0.6931471805599453f * gs_inv);  // (q came pre-scaled by log2(e)/sqrt(D))
    dvp[(size_t)(r + 4 * half) * l + k0 + l31] = dvr * gs_inv;
  }
}

template <int PREC>
static int launch_attention_bwd_mfma8(const float* qkv, const float* o, const float* dout, const float* lse, float* dqkv,
                                      float* dsum, int n, int c, int heads, int l, hipStream_t st) {
  const float qscale = 1.4426950408889634f / sqrtf(8.0f);
  const dim3 grid(cdiv(l, 32 * ATM_NW) * heads * n), block(64 * ATM_NW);
  hipLaunchKernelGGL(attention_bwd_dq_mfma8_kernel<PREC>, grid, block, 0, st, qkv, o, dout, lse, dqkv, dsum, c, heads, l, qscale);
  DSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(attention_bwd_dkv_mfma8_kernel<PREC>, grid, block, 0, st, qkv, dout, lse, dsum, dqkv, c, heads, l, qscale);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

template <int D>
static int launch_attention_bwd(const float* qkv, const float* o, const float* dout, const float* lse, float* dqkv,
                                float* dsum, int n, int c, int heads, int l, hipStream_t st) {
  const float qscale = 1.4426950408889634f / sqrtf((float)D);
  dim3 grid(cdiv(l, 256), heads, n);
  hipLaunchKernelGGL((attention_bwd_dq_kernel<D>), grid, dim3(256), 0, st, qkv, o, dout, lse, dqkv, dsum, c, heads, l,
                     qscale);
  DSG_LAUNCH_CHECK();
  hipLaunchKernelGGL((attention_bwd_dkv_kernel<D>), grid, dim3(256), 0, st, qkv, dout, lse, dsum, dqkv, c, heads, l,
                     qscale);
  DSG_LAUNCH_CHECK();
  return DSG_OK;
}

}  // namespace dsg

// dtype: DSG_F32 = the exact kernels; DSG_BF16 / DSG_F16 = the matrix-core kernels (head_dim 8, l % 32 == 0; anything else: the
// exact ones).  DSG_F16: dO enters the fp16 products scaled by a power of two (kernel comments), whatever the loss scale is.
DSG_API int dsg_attention_bwd_dt(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                                 float* dsum_ws, int32_t n, int32_t c, int32_t heads, int32_t l, int32_t dtype, void* stream) {
  DSG_CHECK_ARG(qkv && out && dout && lse && dqkv && dsum_ws, "dsg_attention_bwd_dt: NULL pointer");
  DSG_CHECK_ARG(n > 0 && c > 0 && heads > 0 && l > 0 && c % heads == 0, "dsg_attention_bwd_dt: bad dims");
  DSG_CHECK_ARG(dtype >= DSG_F32 && dtype <= DSG_F16, "dsg_attention_bwd_dt: bad dtype %d", dtype);
  if (dtype != DSG_F32 && c / heads == 8 && l % 32 == 0 && dsg::g_att_mfma)
    return dtype == DSG_BF16
               ? dsg::launch_attention_bwd_mfma8<1>(qkv, out, dout, lse, dqkv, dsum_ws, n, c, heads, l, static_cast<hipStream_t>(stream))
               : dsg::launch_attention_bwd_mfma8<2>(qkv, out, dout, lse, dqkv, dsum_ws, n, c, heads, l, static_cast<hipStream_t>(stream));
  return dsg_attention_bwd(qkv, out, dout, lse, dqkv, dsum_ws, n, c, heads, l, stream);
}

DSG_API int dsg_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                              float* dsum_ws, int32_t n, int32_t c, int32_t heads, int32_t l, void* stream) {
  DSG_CHECK_ARG(qkv && out && dout && lse && dqkv && dsum_ws, "dsg_attention_bwd: NULL pointer");
  DSG_CHECK_ARG(n > 0 && c > 0 && heads > 0 && l > 0 && c % heads == 0, "dsg_attention_bwd: bad dims");
  DSG_CHECK_ARG(heads <= 65535 && n <= 65535, "dsg_attention_bwd: grid too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  // head_dim 8 (every attention block of the reference's networks): the matrix-core kernels in their fp32-class form (PREC 0);
  // tuning key 14 = 0 keeps the VALU kernels for A/B
  if (c / heads == 8 && l % 32 == 0 && dsg::g_att_mfma && dsg::g_att_bwd_split)
    return dsg::launch_attention_bwd_mfma8<0>(qkv, out, dout, lse, dqkv, dsum_ws, n, c, heads, l, st);
  switch (c / heads) {
    case 8: return dsg::launch_attention_bwd<8>(qkv, out, dout, lse, dqkv, dsum_ws, n, c, heads, l, st);
    case 16: return dsg::launch_attention_bwd<16>(qkv, out, dout, lse, dqkv, dsum_ws, n, c, heads, l, st);
    case 32: return dsg::launch_attention_bwd<32>(qkv, out, dout, lse, dqkv, dsum_ws, n, c, heads, l, st);
    case 64: return dsg::launch_attention_bwd<64>(qkv, out, dout, lse, dqkv, dsum_ws, n, c, heads, l, st);
    default:
      return dsg::fail(DSG_ERR_UNSUPPORTED_SHAPE, "dsg_attention_bwd: head_dim %d not in {8,16,32,64}", c / heads);
  }
}
