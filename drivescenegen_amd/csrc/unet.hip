// Whole-network plan: diffusers' UNet2DModel.forward as one host call that enqueues the fused
// gfx950 kernels of conv.hip / groupnorm.hip / attention.hip / temb.hip on the caller's stream.
//
// Reference: the model is constructed at DriveSceneGen/scripts/train.py:39-57 and evaluated once per
// denoising step inside DDPMPipeline.__call__ (DriveSceneGen/pipeline/training_pipeline.py:26-32,
// DriveSceneGen/scripts/generation.py:14-20).  Block wiring follows SURVEY.md App. A.1 (constructor
// resolution) and A.2 (forward order, skip bookkeeping); parameters are addressed by their diffusers
// state-dict keys (App. A.5) so checkpoints load unchanged.
//
// Data layout in HBM: activations NCHW fp32, carved from ONE caller-owned workspace by a first-fit
// arena whose decisions depend only on (config, batch) -- the same dry run sizes the workspace.
// Concat, nearest-upsample, GroupNorm-apply+SiLU, bias, time-embedding add and residual add never
// materialise: they are folded into the gather / epilogue of dsg_conv2d_fwd.
#include "dsg_common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

namespace dsg {
int conv2d_fwd_impl(const dsg_conv_args* a, hipStream_t st, int force_direct);
int conv_h2_tuning_epoch();
int attention_fwd_exact(const float* qkv, float* out, int n, int c, int heads, int l, hipStream_t st);
bool attention_blocked_ok(int c, int heads, int l);
}

namespace {

using dsg::fail;

struct Arena {
  // free list sorted by offset; capacity is unbounded, the high-water mark is what matters
  std::vector<std::pair<size_t, size_t>> free_;  // (offset, size)
  size_t high = 0;
  Arena() { free_.push_back({0, (size_t)1 << 62}); }
  size_t alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (size_t i = 0; i < free_.size(); ++i) {
      if (free_[i].second >= bytes) {
        const size_t off = free_[i].first;
        free_[i].first += bytes;
        free_[i].second -= bytes;
        if (free_[i].second == 0) free_.erase(free_.begin() + i);
        if (off + bytes > high) high = off + bytes;
        return off;
      }
    }
    return (size_t)-1;
  }
  void release(size_t off, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    size_t i = 0;
    while (i < free_.size() && free_[i].first < off) ++i;
    free_.insert(free_.begin() + i, {off, bytes});
    if (i + 1 < free_.size() && free_[i].first + free_[i].second == free_[i + 1].first) {
      free_[i].second += free_[i + 1].second;
      free_.erase(free_.begin() + i + 1);
    }
    if (i > 0 && free_[i - 1].first + free_[i - 1].second == free_[i].first) {
      free_[i - 1].second += free_[i].second;
      free_.erase(free_.begin() + i);
    }
  }
};

struct Buf {
  Arena* a;
  size_t off, bytes;
  Buf(Arena* a_, size_t o, size_t b) : a(a_), off(o), bytes(b) {}
  ~Buf() { a->release(off, bytes); }
};

struct T {  // activation [N, c, h, w] (or any scratch when c/h/w are unused)
  std::shared_ptr<Buf> b;
  float* p = nullptr;
  int c = 0, h = 0, w = 0;
  int blk = 0;  // 1: channel-blocked [N][c/8][h][w][8] (dsg_conv_args.*_layout); the net's intermediates are
  // per-tile (sum, sum of squares) of every channel, written by the conv that produced the tensor
  std::shared_ptr<Buf> sb;
  double* stats = nullptr;
  int stiles = 0;
  // range guard of the split path: [N] upper bounds of max|x| (float bits), written with the statistics
  unsigned* bound = nullptr;
};

struct Conv {
  float* w = nullptr; float* b = nullptr; int cin = 0, cout = 0, k = 0, wstride = 0;
  void* wh = nullptr; void* whf = nullptr; void* whs = nullptr;
  // range guard: bit i set = the i-th weight tensor packed into this conv has max|w| outside [2^-8, 3e4], where the
  // fp16 pairs of the split stop carrying fp32's precision (or overflow): the conv then runs on the exact f32 MFMA
  unsigned off_split = 0;
};
struct GN { float* g = nullptr; float* b = nullptr; int c = 0; };
struct Res { GN n1; Conv c1; int toff = 0; GN n2; Conv c2; bool sc = false; Conv csc; int cin = 0, cout = 0; };
struct Att { GN gn; Conv qkv; Conv out; int c = 0, heads = 0; };
struct Stage { std::vector<Res> res; std::vector<Att> att; bool resample = false; Conv rconv; };

enum ParamKind { P_COPY, P_CONV };
struct Param {
  std::string name;
  ParamKind kind;
  float* dst;
  int64_t numel;
  int cout, cin, k, cout_total, cout_off;
  bool set;
  void* wh;   // fp16x2-split copy of a 3x3 weight (conv_h2.hip), or nullptr
  void* whf;  // up-sampler convs: the same weight folded into four 2x2 phase kernels, or nullptr
  void* whs;  // down-sampler convs: the same weight over the space-to-depth image (stride 2 on the split path), or nullptr
  Conv* conv = nullptr;  // the conv this weight belongs to (range guard), and its bit in Conv::off_split
  int conv_bit = 0;
};

}  // namespace

namespace dsg {
// intermediates of dsg_unet_forward in the channel-blocked layout (tuning key 13: A/B against [N,C,H,W])
static int g_unet_blocked = 1;
int unet_blocked() { return g_unet_blocked; }
void unet_set_blocked(int v) { g_unet_blocked = v; }
}  // namespace dsg

struct dsg_unet {
  dsg_unet_config cfg;
  std::vector<void*> allocs;
  std::vector<Param> params;
  std::map<std::string, int> index;
  int temb_dim = 0, proj_total = 0;
  float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr, *wp = nullptr, *bp = nullptr;
  float* freqs = nullptr;  // [boc0/2] sinusoid frequencies
  Conv conv_in, conv_out;
  GN norm_out;
  std::vector<Stage> down, up;
  Res mid0, mid1;
  bool mid_attn = false;
  Att mid_att;
  std::map<std::tuple<int, int, int>, size_t> ws_cache;  // (batch, blocked layout, tuning epoch) -> workspace bytes
  std::string err;
  int dt() const { return cfg.compute_dtype; }  // dsg_dtype of the channel-blocked intermediates / matrix-core products
  // range guard of the weights (dsg_unet_set_param): max|w| of every uploaded conv weight lands in wmax_dev[param index],
  // asynchronously; commit_ranges() fetches the table with ONE copy + ONE synchronisation per parameter refresh and turns
  // it into the convs' off_split bits
  float* wmax_dev = nullptr;
  std::vector<float> wmax_host;
  std::vector<int> pending;        // parameter indices whose maxima have not been read back yet
  hipStream_t pending_stream = nullptr;
  int commit_ranges() {
    if (pending.empty()) return DSG_OK;
    wmax_host.resize(params.size());
    if (hipMemcpyAsync(wmax_host.data(), wmax_dev, params.size() * sizeof(float), hipMemcpyDeviceToHost, pending_stream) != hipSuccess ||
        hipStreamSynchronize(pending_stream) != hipSuccess)
      return DSG_ERR_HIP;
    bool changed = false;
    for (int i : pending) {
      Param& p = params[i];
      const float wmax = wmax_host[i];
      const bool bad = !(wmax <= 3.0e4f) || (wmax != 0.f && wmax < 0.00390625f);
      const unsigned before = p.conv->off_split;
      if (bad) p.conv->off_split |= 1u << p.conv_bit;
      else p.conv->off_split &= ~(1u << p.conv_bit);
      changed |= before != p.conv->off_split;
    }
    pending.clear();
    if (changed) ws_cache.clear();  // (a conv that changes kernels may change what the arena holds)
    return DSG_OK;
  }

  ~dsg_unet() {
    for (void* p : allocs)
      if (p) (void)hipFree(p);
  }

  bool alloc_failed = false;   // a failed hipMalloc anywhere in the plan: dsg_unet_create refuses to hand the plan out
  float* dalloc(int64_t numel) {
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)numel * sizeof(float)) != hipSuccess) {
      alloc_failed = true;
      return nullptr;
    }
    allocs.push_back(p);
    return static_cast<float*>(p);
  }
  void add_param(const std::string& name, ParamKind kind, float* dst, int64_t numel, int cout = 0, int cin = 0,
                 int k = 0, int cout_total = 0, int cout_off = 0) {
    index[name] = (int)params.size();
    params.push_back({name, kind, dst, numel, cout, cin, k, cout_total, cout_off, false, nullptr, nullptr, nullptr});
  }
  void reg_conv(const std::string& pre, Conv& c, int cin, int cout, int k, bool upsampler = false,
                bool downsampler = false) {
    c.cin = cin; c.cout = cout; c.k = k;
    c.wstride = (cout + 31) / 32 * 32;  // zero-padded columns: every conv takes the matrix-core path
    c.w = dalloc((int64_t)cin * k * k * c.wstride);
    if (c.w && c.wstride != cout) (void)hipMemset(c.w, 0, (size_t)cin * k * k * c.wstride * sizeof(float));
    c.b = dalloc(cout);
    add_param(pre + ".weight", P_CONV, c.w, (int64_t)cin * k * k * cout, cout, cin, k, c.wstride, 0);
    params.back().conv = &c;
    // second copy, packed for the matrix-core kernel: pre-split fp16 pairs (the same bytes as the fp32 copy), or the
    // weights rounded to the 16-bit compute type (conv_out's 4 / 8 columns are zero-padded to a 64-column tile there)
    if (cin % 16 == 0 && (cout % 64 == 0 || (dt() != DSG_F32 && cout % 8 == 0))) {
      auto packed = [&](int kind) -> void* {
        size_t bytes = 0;
        if (dsg_conv_weight_pack_bytes(cout, cin, k, kind, dt(), 0, &bytes) != DSG_OK) return nullptr;
        float* p = dalloc((int64_t)(bytes + 3) / 4);
        if (p && cout % 64) (void)hipMemset(p, 0, bytes);  // (the padding columns are read by the cout tile)
        return p;
      };
      c.wh = packed(0);
      params.back().wh = c.wh;
      if (downsampler && k == 3) {  // Downsample2D: 2x2 conv over the space-to-depth image (4 cin, 4 of 9 taps)
        c.whs = packed(2);
        params.back().whs = c.whs;
      }
      if (upsampler && k == 3) {  // Upsample2D + conv as four 2x2 convs of the low-resolution map
        c.whf = packed(1);
        params.back().whf = c.whf;
      }
    }
    add_param(pre + ".bias", P_COPY, c.b, cout);
  }
  void reg_gn(const std::string& pre, GN& g, int c) {
    g.c = c;
    g.g = dalloc(c);
    g.b = dalloc(c);
    add_param(pre + ".weight", P_COPY, g.g, c);
    add_param(pre + ".bias", P_COPY, g.b, c);
  }
  void reg_res(const std::string& pre, Res& r, int cin, int cout) {
    r.cin = cin; r.cout = cout;
    reg_gn(pre + ".norm1", r.n1, cin);
    reg_conv(pre + ".conv1", r.c1, cin, cout, 3);
    r.toff = proj_total;
    proj_total += cout;
    reg_gn(pre + ".norm2", r.n2, cout);
    reg_conv(pre + ".conv2", r.c2, cout, cout, 3);
    r.sc = cin != cout;
    if (r.sc) reg_conv(pre + ".conv_shortcut", r.csc, cin, cout, 1);
  }
  void reg_att(const std::string& pre, Att& a, int c) {
    a.c = c;
    a.heads = c / cfg.attention_head_dim;
    reg_gn(pre + ".group_norm", a.gn, c);
    a.qkv.cin = c; a.qkv.cout = 3 * c; a.qkv.k = 1; a.qkv.wstride = 3 * c;
    a.qkv.w = dalloc((int64_t)c * 3 * c);
    a.qkv.b = dalloc(3 * c);
    if (c % 64 == 0) {
      size_t bytes = 0;
      if (dsg_conv_weight_pack_bytes(c, c, 1, 0, dt(), 3 * c, &bytes) == DSG_OK) a.qkv.wh = dalloc((int64_t)(bytes + 3) / 4);
    }
    const char* names[3] = {"to_q", "to_k", "to_v"};
    for (int i = 0; i < 3; ++i) {
      add_param(pre + "." + names[i] + ".weight", P_CONV, a.qkv.w, (int64_t)c * c, c, c, 1, 3 * c, i * c);
      params.back().wh = a.qkv.wh;
      params.back().conv = &a.qkv;
      params.back().conv_bit = i;
      add_param(pre + "." + names[i] + ".bias", P_COPY, a.qkv.b + i * c, c);
    }
    reg_conv(pre + ".to_out.0", a.out, c, c, 1);
  }
};

namespace {

// second pass over the resnets: time_emb_proj rows live in one [proj_total][dim] matrix
void reg_tproj(dsg_unet* h, const std::string& pre, const Res& r) {
  h->add_param(pre + ".time_emb_proj.weight", P_COPY, h->wp + (int64_t)r.toff * h->temb_dim,
               (int64_t)r.cout * h->temb_dim);
  h->add_param(pre + ".time_emb_proj.bias", P_COPY, h->bp + r.toff, r.cout);
}

struct Runner {
  dsg_unet* h;
  int B;
  char* ws;
  bool dry;
  hipStream_t st;
  Arena arena;
  int rc = DSG_OK;
  bool blocked = false;  // intermediates channel-blocked (every channel count a multiple of 8; tuning key 13)
  // range-guard bounds: one zeroed region at the start of the workspace, a slot of B words per tensor
  static constexpr int kBoundSlots = 512;
  unsigned* bound_base = nullptr;
  int bound_used = 0;
  unsigned* bound_slot() {
    if (dt() != DSG_F32 || bound_used >= kBoundSlots) return nullptr;
    return bound_base + (size_t)(bound_used++) * B;
  }

  int dt() const { return h->cfg.compute_dtype; }
  // bytes per element of an activation: channel-blocked intermediates are 16-bit in the mixed-precision modes
  size_t esz_of(int blk) const { return (blk && dt() != DSG_F32) ? 2 : sizeof(float); }

  T alloc(int c, int hh, int w, size_t elems = 0, size_t esz = sizeof(float)) {
    T t;
    const size_t n = elems ? elems : (size_t)B * c * hh * w;
    const size_t bytes = n * esz;
    const size_t off = arena.alloc(bytes);
    t.b = std::make_shared<Buf>(&arena, off, bytes);
    t.p = reinterpret_cast<float*>(ws + off);
    t.c = c; t.h = hh; t.w = w;
    return t;
  }
  bool ok() const { return rc == DSG_OK; }

  // scale/shift of GroupNorm(gn) over cat(x, skip).  Statistics come from the producing convs' epilogues where
  // they wrote them; a tensor without them (conv_in, stride-2 convs, tiny maps) gets a pass of its own.
  // Statistics pass for a tensor whose producer could not write them (conv_in, shapes the split kernels do not
  // take): run once, kept with the tensor -- a skip connection is normalised a second time on the way up.
  void ensure_stats(T& t) {
    if (t.stiles > 0) return;
    const int hw = t.h * t.w;
    // (blocked tensors have only C/8 * N channel blocks to spread over the chip: split the pixels as well)
    int splits = 1;
    if (t.blk)
      while (splits < 16 && hw % (2 * splits) == 0 && hw / (2 * splits) >= 2048) splits *= 2;
    const size_t bytes = (size_t)B * t.c * splits * 2 * sizeof(double);
    const size_t off = arena.alloc(bytes);
    t.sb = std::make_shared<Buf>(&arena, off, bytes);
    t.stats = reinterpret_cast<double*>(ws + off);
    t.stiles = splits;
    if (!dry && ok())
      rc = t.blk ? dsg_gn_channel_stats_blocked_dt(t.p, t.c, B, hw, splits, t.stats, dt(), st)
                 : dsg_gn_channel_stats(t.p, t.c, nullptr, 0, B, hw, t.stats, st);
  }

  // range bound of a tensor that a conv reads WITHOUT a norm in front (up- / down-sampler inputs), from its statistics
  void ensure_bound(T& t) {
    ensure_stats(t);
    if (t.bound) return;
    t.bound = bound_slot();
    if (t.bound && !dry && ok()) rc = dsg_range_bound_from_stats(t.stats, B, t.c, t.stiles, t.bound, st);
  }

  // scale/shift of GroupNorm(gn) over cat(x, skip) from the per-tile statistics both tensors carry
  // want_bound: also leave the range bound of cat(x, skip) in a fresh slot (the resnet's shortcut conv reads them raw)
  T gn_ss(T& x, T* skip, const GN& gn, unsigned** want_bound = nullptr) {
    const int c = x.c + (skip ? skip->c : 0);
    ensure_stats(x);
    if (skip) ensure_stats(*skip);
    T ss = alloc(0, 0, 0, (size_t)B * c * 2);
    unsigned* bnd = want_bound ? bound_slot() : nullptr;
    if (want_bound) *want_bound = bnd;
    if (!dry && ok())
      rc = dsg_gn_finalize_parts_bound(x.stats, x.c, x.stiles, skip ? skip->stats : nullptr, skip ? skip->c : 0,
                                       skip ? skip->stiles : 0, gn.g, gn.b, B, h->cfg.norm_num_groups, x.h * x.w,
                                       h->cfg.norm_eps, ss.p, bnd, st);
    return ss;
  }

  // want_stats: the result feeds a GroupNorm -- have the conv write its per-tile statistics when it can
  // dst_blk: layout of the result (-1: the net's default for intermediates)
  // A resnet's shortcut offered to its conv2 for fusion (dsg_conv_args.sc_*): the 1x1 conv, its source(s) -- the resnet's
  // raw input -- and their range bound.  `taken` tells the caller whether conv2 contracted it.
  struct Shortcut {
    const T* x; const T* skip; const Conv* cv; const unsigned* bound; bool taken;
  };

  T conv(const T& x, const T* skip, const Conv& cv, int stride, int ups, const T* ss, int silu, const float* temb,
         const T* res, float* dst_override = nullptr, bool want_stats = false, int dst_blk = -1,
         const unsigned* raw_bound = nullptr, Shortcut* sc = nullptr) {
    if (dst_blk < 0) dst_blk = blocked ? 1 : 0;
    if (ok() && ((skip && skip->blk != x.blk) || (res && res->blk != dst_blk)))
      rc = dsg::fail(DSG_ERR_INVALID_ARG, "dsg_unet_forward: mixed activation layouts in one conv (internal)");
    const int hc = ups ? 2 * x.h : x.h, wc = ups ? 2 * x.w : x.w;
    const int pad = cv.k / 2;
    const int ho = (hc + 2 * pad - cv.k) / stride + 1, wo = (wc + 2 * pad - cv.k) / stride + 1;
    T y;
    if (dst_override) {
      y.p = dst_override; y.c = cv.cout; y.h = ho; y.w = wo;
    } else {
      y = alloc(cv.cout, ho, wo, 0, esz_of(dst_blk));
    }
    dsg_conv_args a;
    std::memset(&a, 0, sizeof(a));
    a.src0 = x.p; a.c0 = x.c;
    a.src1 = skip ? skip->p : nullptr; a.c1 = skip ? skip->c : 0;
    a.n = B; a.hin = x.h; a.win = x.w; a.upsample = ups; a.ksize = cv.k; a.stride = stride; a.cout = cv.cout;
    a.weight = cv.w; a.weight_cout_stride = cv.wstride; a.bias = cv.b;
    if (!cv.off_split || dt() != DSG_F32) {  // (weights out of the split's range: the exact f32 MFMA kernel serves the conv)
      a.weight_h2 = cv.wh; a.weight_h2_fold = cv.whf; a.weight_h2_s2 = cv.whs;
    }
    if (!ss && raw_bound) {  // un-normalised source(s): hand the kernel their range bound (one for the concatenation)
      a.src_bound = raw_bound;
    } else if (!ss && x.bound && (!skip || skip->bound)) {
      a.src_bound = x.bound;
      a.src_bound1 = skip ? skip->bound : nullptr;
    }
    a.gn_scale_shift = ss ? ss->p : nullptr; a.silu = silu;
    a.temb = temb; a.temb_stride = h->proj_total;
    a.residual = res ? res->p : nullptr;
    a.dst = y.p;
    a.src_layout = x.blk; a.dst_layout = dst_blk;
    a.compute_dtype = dt();
    y.blk = dst_blk;
    T splitk;  // scratch of the small-batch split-K path (lives until the call has been enqueued: stream order)
    if (ok() && dt() == DSG_F32 && !(h->cfg.flags & DSG_UNET_BATCH_INVARIANT)) {
      size_t sk = 0;
      rc = dsg_conv2d_splitk_bytes(&a, &sk);
      if (ok() && sk > 0) {
        splitk = alloc(0, 0, 0, sk / sizeof(float));
        a.splitk_ws = splitk.p;
        a.splitk_ws_bytes = sk;
      }
    }
    if (sc && ok()) {  // offer the shortcut; keep it only if this call's kernel contracts it
      sc->taken = false;
      if (!res && sc->cv->wh && !sc->cv->off_split && !cv.off_split && sc->x->blk == x.blk &&
          (!sc->skip || sc->skip->blk == x.blk)) {
        a.sc_src0 = sc->x->p; a.sc_c0 = sc->x->c;
        a.sc_src1 = sc->skip ? sc->skip->p : nullptr; a.sc_c1 = sc->skip ? sc->skip->c : 0;
        a.sc_weight_h2 = sc->cv->wh; a.sc_bias = sc->cv->b;
        a.sc_src_bound = sc->bound;   // (one bound for the concatenation: gn_ss leaves it)
        int32_t yes = 0;
        rc = dsg_conv2d_fuses_shortcut(&a, &yes);
        sc->taken = ok() && yes != 0;
        if (!sc->taken) {
          a.sc_src0 = a.sc_src1 = a.sc_weight_h2 = nullptr; a.sc_bias = nullptr; a.sc_src_bound = nullptr;
          a.sc_c0 = a.sc_c1 = 0;
        }
      }
      if (!sc->taken) return y;  // (the caller runs the 1x1 on its own and calls again with its result as the residual)
    }
    // pre-staged operand image: where the kernel takes one and the patch would be staged by several cout tiles, the
    // normalise + activate + split work is done once, by a streaming pass, instead of once per cout tile in the K loop
    T operand;  // (lives until the call has been enqueued: stream order)
    if (ok() && dt() == DSG_F32 && x.blk && dst_blk) {
      int32_t yes = 0;
      rc = dsg_conv2d_takes_operand(&a, &yes);
      if (ok() && yes) {
        size_t ob = 0;
        rc = dsg_conv_operand_bytes(B, x.c + (skip ? skip->c : 0), x.h, x.w, DSG_F32, &ob);
        if (ok()) {
          operand = alloc(0, 0, 0, ob / sizeof(float));
          a.src_operand = operand.p;
          if (!dry)
            rc = dsg_conv_operand_prepare(x.p, x.c, skip ? skip->p : nullptr, skip ? skip->c : 0, B, x.h, x.w,
                                          a.gn_scale_shift, a.silu, a.src_bound, a.src_bound1, operand.p, DSG_F32, st);
        }
      }
    }
    if (want_stats && ok()) {
      int32_t tiles = 0;
      rc = dsg_conv2d_stats_tiles(&a, &tiles);
      if (ok() && tiles > 0) {
        const size_t bytes = (size_t)B * cv.cout * tiles * 2 * sizeof(double);
        const size_t off = arena.alloc(bytes);
        y.sb = std::make_shared<Buf>(&arena, off, bytes);
        y.stats = reinterpret_cast<double*>(ws + off);
        y.stiles = tiles;
        a.stats_out = y.stats;
      }
    }
    if (!dry && ok()) rc = dsg::conv2d_fwd_impl(&a, st, 0);
    return y;
  }

  T resnet(T& x, T* skip, const Res& r, const float* tproj) {
    unsigned* raw = nullptr;
    T ss1 = gn_ss(x, skip, r.n1, r.sc ? &raw : nullptr);
    T hmid = conv(x, skip, r.c1, 1, 0, &ss1, 1, tproj + r.toff, nullptr, nullptr, true);
    ss1 = T();
    T ss2 = gn_ss(hmid, nullptr, r.n2);
    T y;
    if (r.sc) {
      // conv_shortcut rides on conv2's K loop where the kernel takes it (ResnetBlock2D: conv_shortcut(input) + conv2(...));
      // otherwise it is a 1x1 call of its own whose result conv2 adds as its residual
      Shortcut fuse{&x, skip, &r.csc, raw, false};
      y = conv(hmid, nullptr, r.c2, 1, 0, &ss2, 1, nullptr, nullptr, nullptr, true, -1, nullptr, &fuse);
      if (!fuse.taken) {
        y = T();
        T sc = conv(x, skip, r.csc, 1, 0, nullptr, 0, nullptr, nullptr, nullptr, false, -1, raw);
        y = conv(hmid, nullptr, r.c2, 1, 0, &ss2, 1, nullptr, &sc, nullptr, true);
      }
    } else {
      y = conv(hmid, nullptr, r.c2, 1, 0, &ss2, 1, nullptr, &x, nullptr, true);
    }
    return y;
  }

  T attention(T& x, const Att& at) {
    T ss = gn_ss(x, nullptr, at.gn);
    // head_dim 8 is one channel block: with blocked intermediates q, k, v and the attention output stay blocked (in the
    // plan's element type) between the three kernels; otherwise they are [N,3C,L] / [N,C,L] fp32
    const bool blk = x.blk && dsg::attention_blocked_ok(x.c, at.heads, x.h * x.w) && !at.qkv.off_split;
    T qkv = conv(x, nullptr, at.qkv, 1, 0, &ss, 0, nullptr, nullptr, nullptr, false, blk ? 1 : 0);
    ss = T();
    T o = blk ? alloc(x.c, x.h, x.w, 0, esz_of(1)) : alloc(x.c, x.h, x.w);
    o.blk = blk ? 1 : 0;
    if (!dry && ok())
      rc = blk ? dsg_attention_fwd_blocked(qkv.p, o.p, B, x.c, at.heads, x.h * x.w, dt(), st)
               : (at.qkv.off_split ? dsg::attention_fwd_exact(qkv.p, o.p, B, x.c, at.heads, x.h * x.w, st)
                                   : dsg_attention_fwd_dt(qkv.p, o.p, B, x.c, at.heads, x.h * x.w, dt(), st));
    qkv = T();
    // (q / k / v weights beyond the split's range make the attention output's range suspect too: its projection then
    //  takes the exact kernel as well -- o has no norm and no statistics to bound it)
    Conv outc = at.out;
    outc.off_split |= at.qkv.off_split;
    return conv(o, nullptr, outc, 1, 0, nullptr, 0, nullptr, &x, nullptr, true);
  }

  int run(const float* xin, const int64_t* t, float* out) {
    const dsg_unet_config& cfg = h->cfg;
    {  // range-guard slots first (their place does not depend on anything else); zeroed once per forward
      const size_t bytes = (size_t)kBoundSlots * B * sizeof(unsigned);
      const size_t off = arena.alloc(bytes);
      bound_base = reinterpret_cast<unsigned*>(ws + off);
      bound_used = 0;
      // (a kernel, not hipMemsetAsync: see dsg::zero_words)
      if (!dry && dt() == DSG_F32 && dsg::zero_words(bound_base, (size_t)kBoundSlots * B, st) != hipSuccess)
        return rc = dsg::fail(DSG_ERR_HIP, "dsg_unet_forward: zero-fill launch failed");
    }
    blocked = dsg::unet_blocked() != 0 || dt() != DSG_F32;  // (the 16-bit modes exist for channel-blocked tensors only)
    for (int i = 0; i < cfg.num_blocks; ++i) blocked = blocked && cfg.block_out_channels[i] % 8 == 0;
    if (dt() != DSG_F32 && !blocked)
      return rc = dsg::fail(DSG_ERR_UNSUPPORTED_SHAPE, "dsg_unet_forward: the bf16 / fp16 modes need block_out_channels %% 8 == 0");
    T act = alloc(0, 0, 0, (size_t)B * h->temb_dim);
    T tproj = alloc(0, 0, 0, (size_t)B * h->proj_total);
    if (!dry) {
      rc = dsg_time_embed_fwd(t, h->freqs, B, cfg.block_out_channels[0], h->temb_dim, h->w1, h->b1, h->w2, h->b2, act.p, st);
      if (ok()) rc = dsg_linear_fwd(act.p, h->wp, h->bp, tproj.p, B, h->temb_dim, h->proj_total, st);
    }
    act = T();
    T x0;
    x0.p = const_cast<float*>(xin); x0.c = cfg.in_channels; x0.h = cfg.sample_h; x0.w = cfg.sample_w;
    T x = conv(x0, nullptr, h->conv_in, 1, 0, nullptr, 0, nullptr, nullptr, nullptr, true);  // (conv_in.hip leaves norm1's statistics)
    std::vector<T> skips;
    ensure_stats(x);  // (before the copy: the skip connection then carries them)
    skips.push_back(x);
    for (auto& d : h->down) {
      for (size_t j = 0; j < d.res.size(); ++j) {
        x = resnet(x, nullptr, d.res[j], tproj.p);
        if (!d.att.empty()) x = attention(x, d.att[j]);
        ensure_stats(x);
        skips.push_back(x);
      }
      if (d.resample) {
        ensure_bound(x);
        x = conv(x, nullptr, d.rconv, 2, 0, nullptr, 0, nullptr, nullptr, nullptr, true);
        ensure_stats(x);
        skips.push_back(x);
      }
    }
    x = resnet(x, nullptr, h->mid0, tproj.p);
    if (h->mid_attn) x = attention(x, h->mid_att);
    x = resnet(x, nullptr, h->mid1, tproj.p);
    for (auto& u : h->up) {
      for (size_t j = 0; j < u.res.size(); ++j) {
        T s = skips.back();
        skips.pop_back();
        x = resnet(x, &s, u.res[j], tproj.p);
        if (!u.att.empty()) x = attention(x, u.att[j]);
      }
      if (u.resample) {
        ensure_bound(x);
        x = conv(x, nullptr, u.rconv, 1, 1, nullptr, 0, nullptr, nullptr, nullptr, true);
      }
    }
    T ssf = gn_ss(x, nullptr, h->norm_out);
    conv(x, nullptr, h->conv_out, 1, 0, &ssf, 1, nullptr, nullptr, out, false, 0);
    return rc;
  }
};

int check_cfg(const dsg_unet_config* c) {
  DSG_CHECK_ARG(c != nullptr, "dsg_unet_create: cfg is NULL");
  DSG_CHECK_ARG(c->num_blocks >= 1 && c->num_blocks <= 8, "dsg_unet_create: num_blocks %d not in [1,8]",
                c->num_blocks);
  DSG_CHECK_ARG(c->in_channels > 0 && c->out_channels > 0 && c->sample_h > 0 && c->sample_w > 0,
                "dsg_unet_create: non-positive channel / size");
  DSG_CHECK_ARG(c->layers_per_block >= 1 && c->norm_num_groups >= 1 && c->attention_head_dim >= 1,
                "dsg_unet_create: bad layers_per_block / norm_num_groups / attention_head_dim");
  for (int i = 0; i < c->num_blocks; ++i) {
    const int ch = c->block_out_channels[i];
    DSG_CHECK_ARG(ch > 0 && ch % c->norm_num_groups == 0,
                  "dsg_unet_create: block_out_channels[%d]=%d not a positive multiple of norm_num_groups=%d", i, ch,
                  c->norm_num_groups);
  }
  DSG_CHECK_ARG((c->flags & ~(uint32_t)DSG_UNET_BATCH_INVARIANT) == 0, "dsg_unet_create: unknown flags 0x%x", c->flags);
  DSG_CHECK_ARG(c->compute_dtype >= DSG_F32 && c->compute_dtype <= DSG_F16,
                "dsg_unet_create: compute_dtype must be DSG_F32, DSG_BF16 or DSG_F16 (got %d)", c->compute_dtype);
  const int f = 1 << (c->num_blocks - 1);
  DSG_CHECK_ARG(c->sample_h % f == 0 && c->sample_w % f == 0,
                "dsg_unet_create: sample size %dx%d not divisible by 2^(num_blocks-1)=%d", c->sample_h, c->sample_w,
                f);
  return DSG_OK;
}

}  // namespace

DSG_API int dsg_unet_create(const dsg_unet_config* cfg, dsg_unet_t** out) {
  DSG_CHECK_ARG(out != nullptr, "dsg_unet_create: out is NULL");
  *out = nullptr;
  int rc = check_cfg(cfg);
  if (rc != DSG_OK) return rc;
  std::unique_ptr<dsg_unet> h(new dsg_unet());
  h->cfg = *cfg;
  const int nb = cfg->num_blocks;
  const int* boc = cfg->block_out_channels;
  const int hd = cfg->attention_head_dim;
  h->temb_dim = 4 * boc[0];

  h->reg_conv("conv_in", h->conv_in, cfg->in_channels, boc[0], 3);
  h->w1 = h->dalloc((int64_t)h->temb_dim * boc[0]);
  h->b1 = h->dalloc(h->temb_dim);
  h->w2 = h->dalloc((int64_t)h->temb_dim * h->temb_dim);
  h->b2 = h->dalloc(h->temb_dim);
  h->add_param("time_embedding.linear_1.weight", P_COPY, h->w1, (int64_t)h->temb_dim * boc[0]);
  h->add_param("time_embedding.linear_1.bias", P_COPY, h->b1, h->temb_dim);
  h->add_param("time_embedding.linear_2.weight", P_COPY, h->w2, (int64_t)h->temb_dim * h->temb_dim);
  h->add_param("time_embedding.linear_2.bias", P_COPY, h->b2, h->temb_dim);

  int out_ch = boc[0];
  h->down.resize(nb);
  for (int i = 0; i < nb; ++i) {
    const int in_ch = out_ch;
    out_ch = boc[i];
    Stage& d = h->down[i];
    const std::string pre = "down_blocks." + std::to_string(i);
    d.res.resize(cfg->layers_per_block);
    if (cfg->down_attn[i]) {
      DSG_CHECK_ARG(out_ch % hd == 0, "dsg_unet_create: channels %d not divisible by attention_head_dim %d", out_ch, hd);
      d.att.resize(cfg->layers_per_block);
    }
    for (int j = 0; j < cfg->layers_per_block; ++j) {
      h->reg_res(pre + ".resnets." + std::to_string(j), d.res[j], j == 0 ? in_ch : out_ch, out_ch);
      if (cfg->down_attn[i]) h->reg_att(pre + ".attentions." + std::to_string(j), d.att[j], out_ch);
    }
    d.resample = i != nb - 1;
    if (d.resample) h->reg_conv(pre + ".downsamplers.0.conv", d.rconv, out_ch, out_ch, 3, false, true);
  }
  h->reg_res("mid_block.resnets.0", h->mid0, boc[nb - 1], boc[nb - 1]);
  h->mid_attn = cfg->add_attention != 0;
  if (h->mid_attn) {
    DSG_CHECK_ARG(boc[nb - 1] % hd == 0, "dsg_unet_create: channels %d not divisible by attention_head_dim %d",
                  boc[nb - 1], hd);
    h->reg_att("mid_block.attentions.0", h->mid_att, boc[nb - 1]);
  }
  h->reg_res("mid_block.resnets.1", h->mid1, boc[nb - 1], boc[nb - 1]);

  h->up.resize(nb);
  out_ch = boc[nb - 1];
  for (int i = 0; i < nb; ++i) {
    const int prev = out_ch;
    out_ch = boc[nb - 1 - i];
    const int in_ch = boc[nb - 1 - std::min(i + 1, nb - 1)];
    Stage& u = h->up[i];
    const std::string pre = "up_blocks." + std::to_string(i);
    const int nl = cfg->layers_per_block + 1;
    u.res.resize(nl);
    if (cfg->up_attn[i]) {
      DSG_CHECK_ARG(out_ch % hd == 0, "dsg_unet_create: channels %d not divisible by attention_head_dim %d", out_ch, hd);
      u.att.resize(nl);
    }
    for (int j = 0; j < nl; ++j) {
      const int skip = (j == nl - 1) ? in_ch : out_ch;
      const int rin = (j == 0) ? prev : out_ch;
      h->reg_res(pre + ".resnets." + std::to_string(j), u.res[j], rin + skip, out_ch);
      if (cfg->up_attn[i]) h->reg_att(pre + ".attentions." + std::to_string(j), u.att[j], out_ch);
    }
    u.resample = i != nb - 1;
    if (u.resample) h->reg_conv(pre + ".upsamplers.0.conv", u.rconv, out_ch, out_ch, 3, true);
  }
  h->reg_gn("conv_norm_out", h->norm_out, boc[0]);
  h->reg_conv("conv_out", h->conv_out, boc[0], cfg->out_channels, 3);

  // time_emb_proj: one [proj_total][temb_dim] matrix
  h->wp = h->dalloc((int64_t)h->proj_total * h->temb_dim);
  h->bp = h->dalloc(h->proj_total);
  for (int i = 0; i < nb; ++i)
    for (size_t j = 0; j < h->down[i].res.size(); ++j)
      reg_tproj(h.get(), "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), h->down[i].res[j]);
  reg_tproj(h.get(), "mid_block.resnets.0", h->mid0);
  reg_tproj(h.get(), "mid_block.resnets.1", h->mid1);
  for (int i = 0; i < nb; ++i)
    for (size_t j = 0; j < h->up[i].res.size(); ++j)
      reg_tproj(h.get(), "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), h->up[i].res[j]);

  {  // default frequency table: exp evaluated in fp64 and rounded once
    const int half = boc[0] / 2;
    std::vector<float> f(half);
    for (int i = 0; i < half; ++i) {
      const float ex = (-9.210340371976184f * (float)i) / (float)half;
      f[i] = (float)std::exp((double)ex);
    }
    h->freqs = h->dalloc(half);
    if (h->freqs) DSG_HIP(hipMemcpy(h->freqs, f.data(), half * sizeof(float), hipMemcpyHostToDevice));
  }
  // (allocated here, not at the first dsg_unet_set_param: that call is documented as legal under stream capture)
  h->wmax_dev = h->dalloc((int64_t)h->params.size());
  if (h->alloc_failed || h->wmax_dev == nullptr || h->freqs == nullptr)
    return fail(DSG_ERR_HIP, "dsg_unet_create: hipMalloc failed (weights, timestep table or range-guard table)");
  for (auto& p : h->params)
    if (p.dst == nullptr) return fail(DSG_ERR_HIP, "dsg_unet_create: hipMalloc failed for %s", p.name.c_str());
  *out = h.release();
  return DSG_OK;
}

DSG_API void dsg_unet_destroy(dsg_unet_t* h) { delete h; }  // (~dsg_unet frees the plan's device memory)

DSG_API int dsg_unet_set_param(dsg_unet_t* h, const char* name, const float* data, int64_t numel, void* stream) {
  DSG_CHECK_ARG(h && name && data, "dsg_unet_set_param: NULL argument");
  std::string key(name);
  // pre-0.18 diffusers attention names
  static const char* legacy[][2] = {{".query.", ".to_q."}, {".key.", ".to_k."}, {".value.", ".to_v."},
                                    {".proj_attn.", ".to_out.0."}};
  for (auto& l : legacy) {
    const size_t pos = key.find(l[0]);
    if (pos != std::string::npos) key.replace(pos, std::strlen(l[0]), l[1]);
  }
  if (key == "time_proj.freqs") {
    DSG_CHECK_ARG(numel == h->cfg.block_out_channels[0] / 2, "dsg_unet_set_param: time_proj.freqs expects %d elements",
                  h->cfg.block_out_channels[0] / 2);
    DSG_HIP(hipMemcpyAsync(h->freqs, data, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
    return DSG_OK;
  }
  auto it = h->index.find(key);
  if (it == h->index.end()) return fail(DSG_ERR_INVALID_ARG, "dsg_unet_set_param: unknown parameter '%s'", name);
  Param& p = h->params[it->second];
  DSG_CHECK_ARG(p.numel == numel, "dsg_unet_set_param: '%s' expects %lld elements, got %lld", name,
                (long long)p.numel, (long long)numel);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p.kind == P_COPY) {
    DSG_HIP(hipMemcpyAsync(p.dst, data, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else {
    int rc = dsg_conv_weight_relayout(data, p.dst, p.cout, p.cin, p.k, p.cout_total, p.cout_off, stream);
    if (rc != DSG_OK) return rc;
    const int dt = h->dt();
    if (p.conv && dt == DSG_F32 && (p.wh || p.whf || p.whs)) {
      // range guard: max|w| decides whether the fp16 pairs of the split can carry this tensor.  The maximum is left in
      // the plan's device table; dsg_unet_commit_params (or the next forward) reads the whole table back at once
      DSG_CHECK_ARG(h->wmax_dev != nullptr, "dsg_unet_set_param: the plan has no range-guard table");
      if (!h->pending.empty() && h->pending_stream != st) {  // (maxima queued on another stream: settle them there first)
        rc = h->commit_ranges();
        if (rc != DSG_OK) return fail(rc, "dsg_unet_set_param: reading the weight maxima back failed");
      }
      rc = dsg_abs_max(data, numel, h->wmax_dev + it->second, stream);
      if (rc != DSG_OK) return rc;
      h->pending.push_back(it->second);
      h->pending_stream = st;
    }
    if (p.wh) {  // (a column window only where the packed matrix is wider than this weight: the fused q/k/v projection)
      const bool window = p.cout_off != 0 || p.cout_total > (p.cout + 63) / 64 * 64;
      rc = dsg_conv_weight_pack(data, p.wh, p.cout, p.cin, p.k, 0, dt, window ? p.cout_total : 0, p.cout_off, stream);
      if (rc != DSG_OK) return rc;
    }
    if (p.whf) {
      rc = dsg_conv_weight_pack(data, p.whf, p.cout, p.cin, 3, 1, dt, 0, 0, stream);
      if (rc != DSG_OK) return rc;
    }
    if (p.whs) {
      rc = dsg_conv_weight_pack(data, p.whs, p.cout, p.cin, 3, 2, dt, 0, 0, stream);
      if (rc != DSG_OK) return rc;
    }
  }
  p.set = true;
  return DSG_OK;
}

DSG_API int dsg_unet_commit_params(dsg_unet_t* h) {
  DSG_CHECK_ARG(h != nullptr, "dsg_unet_commit_params: handle is NULL");
  const int rc = h->commit_ranges();
  if (rc != DSG_OK) return fail(rc, "dsg_unet_commit_params: reading the weight maxima back failed");
  return DSG_OK;
}

DSG_API int dsg_unet_num_params(const dsg_unet_t* h, int64_t* expected_tensors, int64_t* set_tensors,
                                int64_t* total_elements) {
  DSG_CHECK_ARG(h != nullptr, "dsg_unet_num_params: handle is NULL");
  int64_t ns = 0, ne = 0;
  for (auto& p : h->params) {
    ns += p.set ? 1 : 0;
    ne += p.numel;
  }
  if (expected_tensors) *expected_tensors = (int64_t)h->params.size();
  if (set_tensors) *set_tensors = ns;
  if (total_elements) *total_elements = ne;
  return DSG_OK;
}

DSG_API int dsg_unet_param_name(const dsg_unet_t* h, int64_t index, const char** name, int64_t* numel) {
  DSG_CHECK_ARG(h != nullptr, "dsg_unet_param_name: handle is NULL");
  DSG_CHECK_ARG(index >= 0 && index < (int64_t)h->params.size(), "dsg_unet_param_name: index out of range");
  if (name) *name = h->params[index].name.c_str();
  if (numel) *numel = h->params[index].numel;
  return DSG_OK;
}

DSG_API int dsg_unet_workspace_bytes(dsg_unet_t* h, int32_t batch, size_t* bytes) {
  DSG_CHECK_ARG(h && bytes, "dsg_unet_workspace_bytes: NULL argument");
  DSG_CHECK_ARG(batch > 0, "dsg_unet_workspace_bytes: batch must be positive");
  // (the layout and the kernel-selection switches decide which convs write statistics, hence the arena's layout)
  int rc0 = h->commit_ranges();  // (kernel selection, hence the arena, depends on the weights' range bits)
  if (rc0 != DSG_OK) return fail(rc0, "dsg_unet_workspace_bytes: reading the weight maxima back failed");
  const auto key = std::make_tuple((int)batch, dsg::unet_blocked() ? 1 : 0, dsg::conv_h2_tuning_epoch());
  auto it = h->ws_cache.find(key);
  if (it == h->ws_cache.end()) {
    Runner r{h, batch, nullptr, true, nullptr};
    const int rc = r.run(nullptr, nullptr, nullptr);
    if (rc != DSG_OK) return rc;
    it = h->ws_cache.emplace(key, r.arena.high).first;
  }
  *bytes = it->second;
  return DSG_OK;
}

DSG_API int dsg_unet_forward(dsg_unet_t* h, const float* x, const int64_t* timesteps, float* out, int32_t batch,
                             void* workspace, size_t workspace_bytes, void* stream) {
  DSG_CHECK_ARG(h && x && timesteps && out && workspace, "dsg_unet_forward: NULL argument");
  DSG_CHECK_ARG(batch > 0, "dsg_unet_forward: batch must be positive");
  for (auto& p : h->params)
    if (!p.set) return fail(DSG_ERR_NOT_READY, "dsg_unet_forward: parameter '%s' was never set", p.name.c_str());
  size_t need = 0;
  int rc = dsg_unet_workspace_bytes(h, batch, &need);
  if (rc != DSG_OK) return rc;
  if (workspace_bytes < need)
    return fail(DSG_ERR_WORKSPACE_TOO_SMALL, "dsg_unet_forward: workspace %zu bytes < required %zu", workspace_bytes,
                need);
  Runner r{h, batch, static_cast<char*>(workspace), false, static_cast<hipStream_t>(stream)};
  rc = r.run(x, timesteps, out);
  if (rc == DSG_OK && r.arena.high > workspace_bytes)  // (cannot happen while the dry run and the live run agree)
    return fail(DSG_ERR_WORKSPACE_TOO_SMALL, "dsg_unet_forward: the run carved %zu bytes from a %zu-byte workspace",
                r.arena.high, workspace_bytes);
  return rc;
}
