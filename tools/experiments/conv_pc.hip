// Wave-specialised, persistent form of the fp32-equivalent 3x3 conv for the SHALLOW levels (cin <= 128 at 256^2 / 128^2: the
// resnet convs of the network DriveSceneGen/scripts/train.py:39-57 builds, run at training_pipeline.py:84 and inside
// DDPMPipeline.__call__, training_pipeline.py:26-32 / generation.py:14-20).
//
// Same contract, operands, accumulation order and epilogue as conv_h2_kernel<0, 2, 3, 2, 4, *, 3, 64> -- results are bitwise
// the same -- but the work of a tile is dealt out differently.  conv_h2_kernel's four waves each load, activate, split, issue
// MFMAs and store; on these levels a tile is 4-8 K-chunks long, i.e. 8 us of matrix work inside a 32-us life of which 14 us
// are the prologue's and the epilogue's latency chains, and two such workgroups per CU overlap only by chance (0.41 of the
// f16 roof).  Here ONE workgroup of eight waves owns the CU and walks tiles for the whole launch:
//   waves 4-7 (producers): fetch the halo patches THREE chunks ahead into registers, apply GroupNorm affine + SiLU, split
//              into the (hi, scaled lo) fp16 pair and write the operand image of chunk s + 1 into the LDS buffer the consumers
//              are not reading -- across tile boundaries, so a tile has no prologue;
//   waves 0-3 (consumers): one per SIMD beside a producer: MFMAs on chunk s (their only other instructions are fragment
//              reads and the weight DMAs of chunk s + 1 -- no staging arithmetic, no patch loads, so their memory counter
//              sees nothing but their own DMAs and stores), then the tile's epilogue while the producers stage the next
//              tile's first chunks.
// One workgroup barrier per chunk hands the buffers over; VALU (producer) and matrix (consumer) instructions of the two
// waves of a SIMD issue side by side by construction instead of by the luck of two workgroups' phases.
#include "conv_h2_launch.h"

namespace dsg {

#ifndef PC_PRODUCER_PRIO
#define PC_PRODUCER_PRIO 2
#endif
constexpr int PC_NT = 2, PC_NW = 4, PC_BM = 64;  // consumer geometry: 4 waves x 2 rows x 32 columns x 64 output channels

template <int PREC>
struct PcGeom {
  static constexpr int NP = PREC ? 1 : 2;
  using G = H2Geom<PC_NT, 3, PC_NW, 9, PC_BM, NP>;
  static constexpr int BUF = G::BUF_BYTES;                                              // weights | patch | dump slot
  static constexpr int SS_MAX_C = 256;                                                  // channels a scale/shift table holds
  static constexpr int SS_BYTES = SS_MAX_C * 2 * 4;
  static constexpr int RED_FLOATS = PC_NW * (PC_NT / 2) * 2 * PC_BM;
  static constexpr int SV = 16 * (PC_NT / 2) * 2;
  static constexpr int STATS_BYTES = (RED_FLOATS + PC_NW * SV * 64) * 4;
  static constexpr int LDS_BYTES = 2 * BUF + 2 * SS_BYTES + STATS_BYTES;
};

template <int PREC>
__global__ __launch_bounds__(512, 1) void conv_pc_kernel(ConvH2P p) {
  static_assert(PREC == 0, "fp32-equivalent split only (for now)");
  using PG = PcGeom<PREC>;
  using G = typename PG::G;
  constexpr int NP = PG::NP, NT = PC_NT, NW = PC_NW, BM = PC_BM, MTN = BM / 32, TAPS = 9;
  constexpr int PSZ = G::PSZ, PW = G::PW, WHALFS = G::WHALFS, XHALFS = G::XHALFS, BUF = PG::BUF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  auto buf_of = [&](int i) -> unsigned char* { return smem_raw + (i & 1) * BUF; };
  float* const ssl_base = reinterpret_cast<float*>(smem_raw + 2 * BUF);                  // [2][SS_MAX_C * 2]
  float* const red = reinterpret_cast<float*>(smem_raw + 2 * BUF + 2 * PG::SS_BYTES);   // epilogue statistics tables

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int plane = p.hin * p.win;
  const int nq = p.cin / H2_KC;

  // ---- this workgroup's tiles.  XCD k (workgroup id % 8) owns a contiguous eighth of the spatial tiles and its workgroups
  //      walk that eighth side by side (j, j + 32, ...): neighbouring patches -- shared halo rows, the same patch for the
  //      other cout tile -- are in flight behind one L2 at about the same time
  const int nct = p.cout_pad / BM, nsp = p.tiles_x * p.tiles_y * p.n;
  const bool by_xcd = (nsp & 7) == 0 && (gridDim.x & 7) == 0;
  const int t_first = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_step = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int t_lim = by_xcd ? (nsp >> 3) * nct : nsp * nct;
  const int t_base = by_xcd ? (int)(blockIdx.x & 7) * (nsp >> 3) : 0;
  const int ntl = t_first < t_lim ? (t_lim - t_first + t_step - 1) / t_step : 0;          // tiles of this workgroup
  const int nitems = ntl * nq;
  struct Tile { int n, oy0, ox0, m0, ty, tx; };
  auto tile_of = [&](int k) -> Tile {  // (uniform)
    const int m = t_first + t_step * k;
    const int ct = m % nct, sp = t_base + m / nct;
    const int tx = sp % p.tiles_x, r = sp / p.tiles_x, ty = r % p.tiles_y, n = r / p.tiles_y;
    return Tile{n, ty * 8, tx * H2_TW, ct * BM, ty, tx};
  };
  const bool want_stats = p.stats != nullptr;
  if (nitems == 0) return;

  if (wave >= NW) {
    // ===================================== producers =====================================
    const int ptid = tid - 64 * NW;
    const int g2 = (wave - NW) >> 1;  // k-group of the remainder unit
#ifndef PC_NO_SETPRIO
    // the producers' VALU work must win the issue arbitration against the partner's MFMA stream (which needs one slot in eight)
    __builtin_amdgcn_s_setprio(PC_PRODUCER_PRIO);
#endif
    // staging units (as conv_h2_kernel with 256 threads): 0: g = 0, position ptid; 1: g = 1, position ptid;
    // 2: the remainder, g = g2, position 256 + (ptid & 127) where that is inside the 10 x 34 patch
    int upy[3], upx[3], xo1[3], xo2[3];
    bool uvalid[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int g = i == 0 ? 0 : (i == 1 ? 1 : g2);
      const int pos = i < 2 ? ptid : 256 + (ptid & 127);
      uvalid[i] = pos < PSZ;
      const int pc = uvalid[i] ? pos : 0;
      upy[i] = pc / PW;
      upx[i] = pc - upy[i] * PW;
      const int slot = WHALFS + (g * PSZ + pc) * 8;
      xo1[i] = uvalid[i] ? slot : WHALFS + XHALFS;
      xo2[i] = uvalid[i] ? slot + 2 * PSZ * 8 : WHALFS + XHALFS;
    }
    float xr0[3][8], xr1[3][8];   // raw patch values of the two items in flight (two named sets: no run-time register indexing)
    unsigned okm0 = 0u, okm1 = 0u;
    const char* src0b = static_cast<const char*>(p.src0);
    const char* src1b = static_cast<const char*>(p.src1);
    auto load_item = [&](float (&xr)[3][8], unsigned& okm, const Tile& t, int q) {
      unsigned m = 0u;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int g = i == 0 ? 0 : (i == 1 ? 1 : g2);
        const int cb = q * H2_KC + g * 8;  // first channel of the k-group (uniform)
        const char* base = (cb < p.c0) ? src0b + ((size_t)t.n * p.c0 + cb) * plane * 4
                                       : src1b + ((size_t)t.n * p.c1 + (cb - p.c0)) * plane * 4;
        const int gy = t.oy0 - 1 + upy[i], gx = t.ox0 - 1 + upx[i];
        const bool ok = uvalid[i] && gy >= 0 && gy < p.hin && gx >= 0 && gx < p.win;
        const int off = ok ? gy * p.win + gx : 0;
        const float4* sp = reinterpret_cast<const float4*>(base + (size_t)off * 32);
        const float4 lo = sp[0], hi = sp[1];
        xr[i][0] = lo.x; xr[i][1] = lo.y; xr[i][2] = lo.z; xr[i][3] = lo.w;
        xr[i][4] = hi.x; xr[i][5] = hi.y; xr[i][6] = hi.z; xr[i][7] = hi.w;
        m |= ok ? (1u << i) : 0u;
      }
      okm = m;
    };
    // GroupNorm (scale, shift) of image n: global [c][scale | shift] -> LDS per channel PAIR (sc0, sc1, sh0, sh1)
    auto load_table = [&](int par, int n) {
      float* ssl = ssl_base + par * (PG::SS_MAX_C * 2);
      const float* ssg = p.ss + (size_t)n * p.cin * 2;
      for (int idx = ptid; idx < 2 * p.cin; idx += 256) {
        const int c = idx >> 1, which = idx & 1;
        ssl[4 * (c >> 1) + 2 * which + (c & 1)] = ssg[idx];
      }
    };
    typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
    // the arithmetic of conv_h2_kernel's staging pass, with its contractions written out (tests hold the two to the same bits)
    auto commit_item = [&](const float (&xr)[3][8], unsigned okm, int q, int par, unsigned char* buf) {
      const float* ssl = ssl_base + par * (PG::SS_MAX_C * 2);
      _Float16* xb = reinterpret_cast<_Float16*>(buf);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int g = i == 0 ? 0 : (i == 1 ? 1 : g2);
        const float4* ssq = reinterpret_cast<const float4*>(ssl + 2 * (q * H2_KC + g * 8));
        const bool ok = (okm >> i) & 1u;
        unsigned w1[4], w2[4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          const float4 s4 = ssq[jp];
#ifdef PC_ABL_NOSTAGE
          w1[jp] = ok ? __float_as_uint(xr[i][2 * jp] + s4.x) : 0u;
          w2[jp] = ok ? __float_as_uint(xr[i][2 * jp + 1]) : 0u;
          continue;
#endif
          const float a = __builtin_fmaf(xr[i][2 * jp], s4.x, s4.z), b = __builtin_fmaf(xr[i][2 * jp + 1], s4.y, s4.w);
          const float ra = __builtin_amdgcn_rcpf(1.0f + __expf(-a)), rb = __builtin_amdgcn_rcpf(1.0f + __expf(-b));
          const _Float16 a1 = (_Float16)(a * ra), b1 = (_Float16)(b * rb);
          const half2v h = {a1, b1};
          const half2v l = {(_Float16)(__builtin_fmaf(a, ra, -(float)a1) * 2048.0f), (_Float16)(__builtin_fmaf(b, rb, -(float)b1) * 2048.0f)};
          w1[jp] = ok ? __builtin_bit_cast(unsigned, h) : 0u;   // zero padding applies to the ACTIVATED map
          w2[jp] = ok ? __builtin_bit_cast(unsigned, l) : 0u;
        }
        *reinterpret_cast<st_u32x4*>(xb + xo1[i]) = st_u32x4{w1[0], w1[1], w1[2], w1[3]};
        *reinterpret_cast<st_u32x4*>(xb + xo2[i]) = st_u32x4{w2[0], w2[1], w2[2], w2[3]};
      }
    };

    // item s = (tile s / nq, chunk s % nq); running (tile index, chunk) of the items being loaded / committed
    int lk = 0, lq = 0;        // next item to load
    Tile lt = tile_of(0);
    auto advance_load = [&]() {
      if (++lq == nq) {
        lq = 0;
        ++lk;
        if (lk < ntl) lt = tile_of(lk);
      }
    };
    int tab_par = 0;
    load_table(0, lt.n);
    load_item(xr0, okm0, lt, 0);
    advance_load();
    if (nitems > 1) {
      load_item(xr1, okm1, lt, lq);
      advance_load();
    }
    __syncthreads();  // [B_tab]  the first table is in LDS
    commit_item(xr0, okm0, 0, 0, buf_of(0));
    if (nitems > 2) {
      load_item(xr0, okm0, lt, lq);   // item 2 into the set item 0 has just left
      advance_load();
    }
    __syncthreads();  // [B_init] item 0 is staged
    // commit side: iteration s stages item s + 1 = (tile ck, chunk cq) of image cn
    int ck = 0, cq = 1, cn = tile_of(0).n;
    auto iteration = [&](int s, float (&xr)[3][8], unsigned& okm) {  // (xr: the set that holds item s + 1)
      if (s + 1 < nitems) commit_item(xr, okm, cq, tab_par, buf_of((s + 1) & 1));
      // item s + 2: when it belongs to another image, that image's table goes into the other buffer now -- visible to the
      // item's commit (next iteration) after this iteration's barrier
      int nk = ck, nqq = cq + 1, nn = cn;
      if (nqq == nq) {
        nqq = 0;
        nk = ck + 1;
        if (nk < ntl) nn = tile_of(nk).n;
      }
      const bool flip = s + 2 < nitems && nn != cn;
      if (flip) load_table(tab_par ^ 1, nn);
      if (s + 3 < nitems) {
        load_item(xr, okm, lt, lq);   // item s + 3 into the set item s + 1 has just left
        advance_load();
      }
      __syncthreads();  // [B_s]
      if (flip) tab_par ^= 1;
      ck = nk; cq = nqq; cn = nn;
    };
    for (int s = 0; s < nitems; s += 2) {
      iteration(s, xr1, okm1);
      if (s + 1 < nitems) iteration(s + 1, xr0, okm0);
    }
    if (want_stats) __syncthreads();  // [B_last] the consumers' last statistics tile
    return;
  }

  // ===================================== consumers =====================================
  // weight slab of an item: 36 (piece, tap, g) segments of 64 couts x 16 B, global -> LDS by DMA; wave w moves w, w + 4, ...
  const unsigned segb = (unsigned)p.wh_stride * 16u;
  const unsigned chunkb = G::NSEG * segb;
  int segoff[G::NDMA];
#pragma unroll
  for (int k = 0; k < G::NDMA; ++k) segoff[k] = __builtin_amdgcn_readfirstlane(min(wave + NW * k, G::NUNIT - 1) * (int)segb);
  const int lane16 = lane * 16;
  auto dma_weights = [&](int k, const char* wq, unsigned char* buf) {
    const int unit = min(wave + NW * k, G::NUNIT - 1);
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(buf + unit * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n" ::"v"(lane16), "s"(sgpr_ptr(wq + segoff[k])),
                 "s"(__builtin_amdgcn_readfirstlane(lds_addr))
                 : "memory");
  };
  auto wtile_of = [&](const Tile& t) -> const char* { return static_cast<const char*>(p.wh) + (size_t)t.m0 * 16; };

  {  // item 0's weights
    const Tile t0 = tile_of(0);
#pragma unroll
    for (int d = 0; d < G::NDMA; ++d) dma_weights(d, wtile_of(t0), buf_of(0));
  }
  __syncthreads();  // [B_tab]
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // [B_init]

  // ---- epilogue constants that do not depend on the tile
  constexpr int SV = PG::SV;
  float* stab = red + PG::RED_FLOATS + wave * (SV * 64);
  const int stab_w = half * 32 + (l31 & 3);
  int stab_sw[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) stab_sw[c] = stab_w + (((((l31 >> 2) ^ half) ^ (2 * c)) & 7) << 2);
  const int oplane = p.hout * p.wout;
  const int oplane4 = __builtin_amdgcn_readfirstlane(oplane * 4);

  // statistics tiles are 8 rows x 32 columns = one workgroup tile: the four waves' row pairs, summed in row order in fp64
  auto stats_combine = [&](const Tile& t) {
    if (tid < 2 * BM) {
      const int cl = tid & (BM - 1), which = (tid / BM) & 1;
      if (t.m0 + cl < p.cout) {
        const int ntile = p.tiles_x * p.tiles_y;
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) v += (double)red[(j * 2 + which) * BM + cl];
        const int tile8 = t.ty * p.tiles_x + t.tx;
        p.stats[(((size_t)t.n * p.cout + t.m0 + cl) * ntile + tile8) * 2 + which] = v;
      }
    }
  };
  int s = 0;
  for (int k = 0; k < ntl; ++k) {  // (body not re-indented)
  const Tile tl = tile_of(k);
  const Tile tnext = k + 1 < ntl ? tile_of(k + 1) : tl;
  f32x16 acc_hi[MTN][NT], acc_lo[MTN][NT];
#pragma unroll
  for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_hi[mt][nt][r] = acc_lo[mt][nt][r] = 0.f;
  for (int q = 0; q < nq; ++q, ++s) {
    unsigned char* cur = buf_of(s & 1);
    unsigned char* nxt = buf_of((s + 1) & 1);
    // the next item's weights (its tile's cout tile, its chunk)
    const int nqq = q + 1 == nq ? 0 : q + 1;
    const Tile& ntile = q + 1 == nq ? tnext : tl;
    // (issued unconditionally -- the launch's very last item re-fetches chunk 0 of its own tile into the buffer nobody reads
    //  any more: a branch around the DMAs would fence the scheduler between every tap's reads and MFMAs)
    const char* wqn = wtile_of(ntile) + (size_t)nqq * chunkb;
    const _Float16* wl = reinterpret_cast<const _Float16*>(cur);
    const _Float16* xl = reinterpret_cast<const _Float16*>(cur) + WHALFS;
    half8 fa[2][MTN][NP], fb[2][NT][NP];
    auto load_frags = [&](int tap, int par) {
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
          fa[par][mt][pc] = *reinterpret_cast<const half8*>(wl + (((pc * TAPS + tap) * 2 + half) * BM + mt * 32 + l31) * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
          fb[par][nt][pc] = *reinterpret_cast<const half8*>(xl + ((pc * 2 + half) * PSZ + (wave * NT + nt + dy) * PW + l31 + dx) * 8);
    };
    load_frags(0, 0);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 1 < TAPS) load_frags(tap + 1, (tap + 1) & 1);
      if (tap < TAPS - 1) {
#pragma unroll
        for (int d = tap * G::NDMA / (TAPS - 1); d < (tap + 1) * G::NDMA / (TAPS - 1); ++d) dma_weights(d, wqn, nxt);
      }
      const int par = tap & 1;
#pragma unroll
      for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#ifdef PC_ABL_NOMMA   // (tools/ timing experiments only: wrong results)
          asm volatile("" ::"v"(fa[par][mt][0]), "v"(fa[par][mt][1]), "v"(fb[par][nt][0]), "v"(fb[par][nt][1]));
#else
          acc_hi[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][mt][0], fb[par][nt][0], acc_hi[mt][nt], 0, 0, 0);
          acc_lo[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][mt][0], fb[par][nt][1], acc_lo[mt][nt], 0, 0, 0);
          acc_lo[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[par][mt][1], fb[par][nt][0], acc_lo[mt][nt], 0, 0, 0);
#endif
        }
      // pin the issue order: the next tap's fragment reads go out IN FRONT of this tap's MFMAs, a whole tap (384 matrix
      // cycles) ahead of their use (the scheduler left to itself sinks them to the end of the tap and the next tap's first
      // MFMAs wait for the LDS round trip; one read behind each MFMA was three MFMAs of slack: still waits)
      if (tap + 1 < TAPS) {
        __builtin_amdgcn_sched_group_barrier(0x100, (MTN + NT) * NP, 0);              // the next tap's fragment reads ...
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * MTN * NT, 0);                      // ... then this tap's MFMAs
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's weight DMAs (and the previous tile's stores) are done
    __syncthreads();  // [B_s]
    if (want_stats && q == 0 && k > 0) stats_combine(tile_of(k - 1));  // the previous tile's partials: every wave has left them
  }
#ifdef PC_ABL_NOEPI
    if (p.n < 0)
#endif
    {
      // ---- epilogue of tile tl (conv_h2_kernel's, for this geometry: channel-blocked fp32 dst / residual)
      const bool has_r = p.res != nullptr;
      const int nvalid = min(BM, p.cout - tl.m0);
      const size_t tile_off = ((size_t)tl.n * p.cout + tl.m0) * oplane * 4;
      const int range = nvalid * oplane * 4;
      char* dstb = static_cast<char*>(p.dst);
      const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(dstb + tile_off, 0, range, 0x00020000);
      const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(
          has_r ? const_cast<char*>(static_cast<const char*>(p.res)) + tile_off : dstb, 0, has_r ? range : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t bias_rs = __builtin_amdgcn_make_buffer_rsrc(
          p.bias ? const_cast<float*>(p.bias + tl.m0) : reinterpret_cast<float*>(dstb), 0, p.bias ? nvalid * 4 : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t temb_rs = __builtin_amdgcn_make_buffer_rsrc(
          p.temb ? const_cast<float*>(p.temb + (size_t)tl.n * p.temb_stride + tl.m0) : reinterpret_cast<float*>(dstb), 0,
          p.temb ? nvalid * 4 : 0, 0x00020000);
      int voff[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int oy = tl.oy0 + wave * NT + nt, ox = tl.ox0 + l31;
        voff[nt] = ((oy * p.wout + ox) * 8 + 4 * half) * 4;
      }
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
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const float4 qv = __builtin_bit_cast(
                  float4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, voff[nt], (mt * 4 + rg) * 8 * oplane4, 0));
              rv[4 * rg][nt] = qv.x; rv[4 * rg + 1][nt] = qv.y; rv[4 * rg + 2][nt] = qv.z; rv[4 * rg + 3][nt] = qv.w;
            }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) rv[r][nt] = 0.f;
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          float vv[4][NT];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int r = 4 * rg + j;
              vv[j][nt] = ((acc_hi[mt][nt][r] + acc_lo[mt][nt][r] * (1.0f / 2048.0f)) + addv[r]) + rv[r][nt];
            }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const float4 o = make_float4(vv[0][nt], vv[1][nt], vv[2][nt], vv[3][nt]);
            // (the channel-block offset in the VECTOR offset: see conv_h2_kernel's epilogue)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), dst_rs, voff[nt] + (mt * 4 + rg) * 8 * oplane4, 0, 0);
          }
          if (want_stats) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float a = vv[j][0], b = vv[j][1];
              const int v = (rg * 4 + j) * 2;
              stab[stab_sw[v & 3] + v * 64] = a + b;
              stab[stab_sw[(v + 1) & 3] + (v + 1) * 64] = __builtin_fmaf(a, a, b * b);
            }
          }
        }
        if (want_stats) {
          __builtin_amdgcn_wave_barrier();
          {
            const int row = lane;  // 2 SV = 64 rows: one per lane
            const int v = row >> 1, hh = row & 1;
            const float4* rp = reinterpret_cast<const float4*>(stab + row * 32);
            float t = 0.f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
              const float4 q4 = rp[kk ^ (row & 7)];
              t = (((t + q4.x) + q4.y) + q4.z) + q4.w;
            }
            const int which = v & 1, cj = v >> 1;  // cj = rg * 4 + j
            const int crel = mt * 32 + (cj & 3) + 8 * (cj >> 2);
            red[(wave * 2 + which) * BM + crel + 4 * hh] = t;
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      // (the four waves' row-pair partials are combined behind the NEXT chunk barrier -- stats_combine below -- so the tile
      //  needs no workgroup barrier of its own: the consumers would wait there for the producers' staging work)
    }
  }  // tiles
  if (want_stats) {
    __syncthreads();  // [B_last]
    stats_combine(tile_of(ntl - 1));
  }
}

// the calls this kernel takes (a subset of the two-workgroup kernel's): fp32-equivalent, every tensor channel-blocked, a
// resnet's conv1 / conv2 without the fused shortcut, cin <= 128, a grid that gives every CU several tiles
bool conv_pc_eligible(const dsg_conv_args* a, int hout, int wout, int slices) {
  if (!g_h2.pc || !g_h2.enabled || a->compute_dtype != DSG_F32) return false;
  if (a->ksize != 3 || a->stride != 1 || a->upsample || a->pool2 || a->src_layout != 1 || a->dst_layout != 1) return false;
  if (!a->gn_scale_shift || !a->silu || a->sc_weight_h2 || a->src_operand || a->weight_h2 == nullptr || a->weight_h2_cout_stride) return false;
  const int cin = a->c0 + a->c1;
  if (cin % 16 || cin < 32 || cin > 128 || (a->c1 && a->c0 % 16) || a->cout % 8 || slices != 1) return false;
  if (wout % H2_TW || hout % 8) return false;
  const int cp = (a->cout + 63) / 64 * 64;
  return (wout / H2_TW) * (hout / 8) * a->n * (cp / 64) >= 4 * H2_CUS;
}

int conv_pc_launch(const ConvH2P& p0, hipStream_t st) {
  ConvH2P p = p0;
  using PG = PcGeom<0>;
  static bool raised = false;
  if (!raised) {
    DSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pc_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised = true;
  }
  hipLaunchKernelGGL(conv_pc_kernel<0>, dim3(H2_CUS), dim3(512), (size_t)PG::LDS_BYTES, st, p);
  return DSG_OK;
}

}  // namespace dsg
