"""Thin Python wrappers over the per-op C-ABI entry points of libdsg.so (include/dsg.h).

Used by the training path (autograd.py) and by the parity tests; the sampler hot loop goes through
the whole-network plan (dsg_unet_forward) instead.  GPU tensors only -- no fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib


def _st(t):
    return _lib.stream_ptr(t.device)


def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def relayout_conv_weight(w_oihw: torch.Tensor, out: torch.Tensor = None, cout_total: int = None,
                         cout_off: int = 0) -> torch.Tensor:
    """OIHW (or Linear [out,in]) -> engine layout [Cin][k*k][cout_total]; cout_total defaults to cout rounded up
    to 32 (zero columns) so that every conv can take the matrix-core path."""
    w = w_oihw.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    k = w.shape[2] if w.dim() == 4 else 1
    cout_total = cout_total or _pad32(cout)
    if out is None:
        out = torch.zeros((cin, k * k, cout_total), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().dsg_conv_weight_relayout(_lib.ptr(w), _lib.ptr(out), cout, cin, k, cout_total,
                                                        cout_off, _st(w)))
    return out


_DT_OF = {torch.float32: _lib.DSG_F32, torch.bfloat16: _lib.DSG_BF16, torch.float16: _lib.DSG_F16}
PACK_FWD, PACK_FOLD, PACK_S2, PACK_DGRAD, PACK_DGRAD_S2, PACK_DGRAD_UPS = 0, 1, 2, 3, 4, 5


def dtype_code(dtype) -> int:
    """"fp32" | "bf16" | "fp16" | torch dtype | dsg_dtype code -> dsg_dtype code (include/dsg.h)."""
    if isinstance(dtype, int):
        return dtype
    if dtype in _DT_OF:
        return _DT_OF[dtype]
    return _lib.DTYPE_CODES[dtype or "fp32"]


def pack_conv_weight(w_oihw: torch.Tensor, kind: int = PACK_FWD, dtype=0, n_total: int = 0, n_off: int = 0,
                     out: torch.Tensor = None) -> torch.Tensor:
    """dsg_conv_weight_pack: OIHW fp32 -> the matrix-core operand image for `kind` (PACK_*) in dsg_dtype `dtype`
    (DSG_F32: (hi, scaled lo) fp16 pairs of the fp32-equivalent split; DSG_BF16 / DSG_F16: rounded once).
    Returns a flat int16 buffer (the layout is the kernel's, see include/dsg.h)."""
    w = w_oihw.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    k = w.shape[2] if w.dim() == 4 else 1
    dt = dtype_code(dtype)
    lib = _lib.load()
    if out is None:
        nbytes = C.c_size_t()
        _lib.check(lib.dsg_conv_weight_pack_bytes(cout, cin, k, kind, dt, n_total, C.byref(nbytes)))
        out = torch.zeros(nbytes.value // 2, dtype=torch.int16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.dsg_conv_weight_pack(_lib.ptr(w), out.data_ptr(), cout, cin, k, kind, dt, n_total, n_off, _st(w)))
    return out


class PackTable:
    """A device-resident table of weight-refresh jobs for dsg_conv_weight_pack_batch: every job is one
    `pack_conv_weight(w, kind, dtype, n_total, n_off, out=dst)` call -- or, kind < 0, a copy of w's elements into dst --
    and `run()` refreshes all of them in ONE launch (a training step re-packs every conv weight after optimizer.step()).
    jobs: list of dict(w=, dst=, kind=, dtype=, n_total=0, n_off=0); the tensors must stay where they are."""

    def __init__(self, jobs):
        lib = _lib.load()
        n = len(jobs)
        arr = (_lib.PackJob * n)()
        first = [0]
        for j, job in zip(arr, jobs):
            w, dst, kind = job["w"], job["dst"], int(job["kind"])
            j.w, j.dst, j.kind = w.data_ptr(), dst.data_ptr(), kind
            if kind < 0:
                j.cout = w.numel()
            else:
                j.cout, j.cin = w.shape[0], w.shape[1]
                j.ksize = w.shape[2] if w.dim() == 4 else 1
                j.dtype, j.n_total, j.n_off = dtype_code(job.get("dtype", 0)), int(job.get("n_total", 0)), int(job.get("n_off", 0))
                ndim = j.cin if kind in (PACK_DGRAD, PACK_DGRAD_S2, PACK_DGRAD_UPS) else j.cout
                j.n_pad = ((j.n_total or ndim) + 63) // 64 * 64
            items = C.c_int64()
            _lib.check(lib.dsg_conv_weight_pack_batch_items(C.byref(j), C.byref(items)))
            first.append(first[-1] + items.value)
        dev = jobs[0]["w"].device
        self.n, self.total = n, first[-1]
        self.jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.first = torch.tensor(first, dtype=torch.int64).to(dev)
        self.ptrs = tuple((job["w"].data_ptr(), job["dst"].data_ptr()) for job in jobs)

    def run(self):
        with torch.cuda.device(self.jobs.device):
            _lib.check(_lib.load().dsg_conv_weight_pack_batch(self.jobs.data_ptr(), self.first.data_ptr(), self.n, self.total,
                                                              _st(self.jobs)))


def relayout_conv_weight_h2(w_oihw: torch.Tensor, out: torch.Tensor = None, cout_total: int = None,
                            cout_off: int = 0) -> torch.Tensor:
    """OIHW 3x3 / 1x1 / Linear (cin % 16 == 0) -> fp16x2-split engine layout [Cin/16][2][k*k][2][cout_pad64][8]."""
    w = w_oihw.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    k = w.shape[2] if w.dim() == 4 else 1
    cout_total = cout_total or cout
    if out is None:
        out = torch.zeros((cin // 16, 2, k * k, 2, (cout_total + 63) // 64 * 64, 8), dtype=torch.float16,
                          device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().dsg_conv_weight_relayout_h2(_lib.ptr(w), out.data_ptr(), cout, cin, k, cout_total,
                                                           cout_off, _st(w)))
    return out


def relayout_conv_weight_h2_fold(w_oihw: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """OIHW 3x3 -> the up-sampler conv's folded weights [4 phases][Cin/16][2][2x2][2][cout_pad64][8] fp16."""
    w = w_oihw.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    if out is None:
        out = torch.zeros((4, cin // 16, 2, 4, 2, (cout + 63) // 64 * 64, 8), dtype=torch.float16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().dsg_conv_weight_relayout_h2_fold(_lib.ptr(w), out.data_ptr(), cout, cin, _st(w)))
    return out


def relayout_conv_weight_h2_s2(w_oihw: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """OIHW 3x3 (cin % 8 == 0) -> the stride-2 conv's weights over the space-to-depth image
    [4 Cin/16][2][2x2][2][cout_pad64][8] fp16 (dsg_conv_args.weight_h2_s2)."""
    w = w_oihw.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    if out is None:
        out = torch.zeros((4 * cin // 16, 2, 4, 2, (cout + 63) // 64 * 64, 8), dtype=torch.float16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().dsg_conv_weight_relayout_h2_s2(_lib.ptr(w), out.data_ptr(), cout, cin, _st(w)))
    return out


def relayout_conv_weight_h2_dgrad(w_oihw: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """OIHW (cout % 16 == 0) -> fp16x2-split layout of the data-gradient conv: [Cout/16][2][k*k][2][cin_pad64][8]."""
    w = w_oihw.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    k = w.shape[2] if w.dim() == 4 else 1
    if out is None:
        out = torch.zeros((cout // 16, 2, k * k, 2, (cin + 63) // 64 * 64, 8), dtype=torch.float16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().dsg_conv_weight_relayout_h2_dgrad(_lib.ptr(w), out.data_ptr(), cout, cin, k, _st(w)))
    return out


def relayout_conv_weight_dgrad(w_oihw: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """OIHW -> [Cout][k*k flipped][Cin padded to 32]: weight of the data-gradient conv dX = conv(dY, .)."""
    w = w_oihw.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    k = w.shape[2] if w.dim() == 4 else 1
    if out is None:
        out = torch.zeros((cout, k * k, _pad32(cin)), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().dsg_conv_weight_relayout_dgrad(_lib.ptr(w), _lib.ptr(out), cout, cin, k,
                                                              out.shape[-1], 0, _st(w)))
    return out


def conv2d_fused(src0, weight_r, bias=None, src1=None, ksize=3, stride=1, upsample=False, gn_scale_shift=None,
                 silu=False, temb=None, temb_stride=0, residual=None, out=None, direct=False, cout=None,
                 pool2=False, wstride=None, weight_h2=None, want_stats=False, stats_buf=None, weight_h2_col=0, weight_h2_fold=None,
                 src_blocked=False, dst_blocked=False, weight_h2_s2=None, compute_dtype=0, weight_h2_stride=0,
                 src_bound=None, src_bound1=None, splitk=False, shortcut=None, operand=None, gnb=None, s2_window4=False):
    """dsg_conv2d_fwd: see include/dsg.h.  `weight_r` is in engine layout; `temb` is a [N, temb_stride] view
    whose first `cout` columns (from its data pointer) are added per (n, cout).
    want_stats: also return the per-tile GroupNorm statistics [N][cout][tiles][2] (fp64) of the result, or None
    when the kernel serving the call does not produce them.
    src_blocked / dst_blocked: the sources / (result, residual) are channel-blocked [N, C/8, H, W, 8] tensors.
    compute_dtype: dsg_dtype ("bf16" / "fp16" / code): the channel-blocked tensors are then torch.bfloat16 / float16 and
    weight_h2* must come from pack_conv_weight(..., dtype=the same); [N, C, H, W] tensors stay fp32.
    shortcut: dict(src0=, src1=None, weight_h2=, bias=None, bound=None, bound1=None) -- the resnet's 1x1 conv_shortcut over
    its raw input, contracted in the same kernel (dsg_conv_args.sc_*); raises when the call cannot fuse it
    (ask `conv2d_fuses_shortcut` first).
    operand: None | "auto" | "query" | a tensor from `conv_operand_prepare` -- the pre-staged operand image of the call's
    sources (dsg_conv_args.src_operand).  "auto" asks dsg_conv2d_takes_operand and prepares the image when the answer is
    yes; "query" only returns that answer.
    gnb: dict(x0=, x1=None, ss=, silu=True[, query_only=True]) -- this call is the DATA GRADIENT of a conv behind
    silu?(GroupNorm(cat(x0, x1))): the kernel's epilogue also leaves the norm's backward statistics, per-tile (sum du, sum du * x),
    in the table `want_stats` returns (dsg_conv_args.gnb_*; hand it to gn_bwd* as `parts`).  query_only: just ask
    dsg_conv2d_gnb_supported.  Raises when the call's kernel has no such epilogue.
    s2_window4: with stride=2 and weight_h2_s2 = pack_conv_weight(w, PACK_DGRAD_UPS): the call is the data gradient of the
    up-sampler conv `w` (src0 = dY at full resolution, result = dX at half of it; dsg_conv_args.s2_window4)."""
    lib = _lib.load()
    cdt = dtype_code(compute_dtype)
    blk_dtype = _lib.TORCH_DTYPES[cdt]
    for t, blocked in ((src0, src_blocked), (src1, src_blocked), (residual, dst_blocked), (out, dst_blocked)):
        if t is not None and t.dtype != (blk_dtype if blocked else torch.float32):
            raise RuntimeError(f"conv2d_fused: tensor dtype {t.dtype} does not match compute_dtype / layout "
                               f"(channel-blocked: {blk_dtype}, [N,C,H,W]: float32)")
    if src_blocked:
        n, cb0, hin, win, _ = src0.shape
        c0 = 8 * cb0
        c1 = 8 * src1.shape[1] if src1 is not None else 0
    else:
        n, c0, hin, win = src0.shape
        c1 = src1.shape[1] if src1 is not None else 0
    if weight_r is None:  # allowed when the weight_h2* kernels serve the call (the library says so if they do not)
        if not cout:
            raise RuntimeError("conv2d_fused: cout is required when weight_r is None")
        wptr, wstride = None, wstride or cout
    else:
        wptr = weight_r.data_ptr() if wstride else _lib.ptr(weight_r)  # a column window of a wider matrix is allowed
        wstride = wstride or weight_r.shape[-1]
    cout = cout or wstride
    hc, wc = (2 * hin, 2 * win) if upsample else (hin, win)
    pad = ksize // 2
    ho = (hc + 2 * pad - ksize) // stride + 1
    wo = (wc + 2 * pad - ksize) // stride + 1
    query = (operand == "query" or (shortcut is not None and bool(shortcut.get("query_only")))
             or (gnb is not None and bool(gnb.get("query_only"))))  # host-only: nothing allocated
    if out is None and not query:
        shape = (n, cout, ho // 2, wo // 2) if pool2 else (n, cout, ho, wo)
        if dst_blocked:
            shape = (n, cout // 8, shape[2], shape[3], 8)
        out = torch.empty(shape, dtype=blk_dtype if dst_blocked else torch.float32, device=src0.device)
    a = _lib.ConvArgs()
    a.compute_dtype = cdt
    a.src_layout, a.dst_layout = int(src_blocked), int(dst_blocked)
    a.src0, a.src1 = _lib.ptr(src0), _lib.ptr(src1)
    a.c0, a.c1, a.n, a.hin, a.win = c0, c1, n, hin, win
    a.upsample, a.ksize, a.stride, a.cout = int(upsample), ksize, stride, cout
    a.weight, a.bias = wptr, _lib.ptr(bias)
    a.weight_cout_stride, a.pool2 = wstride, int(pool2)
    if weight_h2 is not None:   # [K/16][pieces][taps][2][cout_total_pad][8] halfs; weight_h2_col selects a column window
        a.weight_h2 = weight_h2.data_ptr() + 16 * int(weight_h2_col)
        if weight_h2_stride:    # flat buffers from pack_conv_weight: the row length is given
            a.weight_h2_cout_stride = int(weight_h2_stride)
        elif weight_h2.dim() >= 2:
            a.weight_h2_cout_stride = weight_h2.shape[-2] if weight_h2_col or weight_h2.shape[-2] != (cout + 63) // 64 * 64 else 0
    a.weight_h2_fold = weight_h2_fold.data_ptr() if weight_h2_fold is not None else None
    a.weight_h2_s2 = weight_h2_s2.data_ptr() if weight_h2_s2 is not None else None
    a.s2_window4 = int(bool(s2_window4))
    a.gn_scale_shift, a.silu = _lib.ptr(gn_scale_shift), int(silu)
    if temb is not None:
        if not temb.is_cuda:
            raise RuntimeError("temb must be a GPU tensor")
        a.temb, a.temb_stride = temb.data_ptr(), int(temb_stride or temb.stride(0))
    a.residual, a.dst = _lib.ptr(residual), _lib.ptr(out)
    # range guard of the split path (include/dsg.h): int32 [N] tensors holding float bits
    a.src_bound, a.src_bound1 = _lib.ptr(src_bound), _lib.ptr(src_bound1)
    if shortcut is not None:
        s0, s1 = shortcut["src0"], shortcut.get("src1")
        a.sc_src0, a.sc_src1 = _lib.ptr(s0), _lib.ptr(s1)
        a.sc_c0 = 8 * s0.shape[1] if src_blocked else s0.shape[1]
        a.sc_c1 = 0 if s1 is None else (8 * s1.shape[1] if src_blocked else s1.shape[1])
        a.sc_weight_h2 = shortcut["weight_h2"].data_ptr()
        a.sc_bias = _lib.ptr(shortcut.get("bias"))
        a.sc_src_bound, a.sc_src_bound1 = _lib.ptr(shortcut.get("bound")), _lib.ptr(shortcut.get("bound1"))
    if gnb is not None:
        gx0, gx1 = gnb["x0"], gnb.get("x1")
        a.gnb_x0, a.gnb_x1 = _lib.ptr(gx0), _lib.ptr(gx1)
        a.gnb_c0 = (8 * gx0.shape[1] if gx0.dim() == 5 else gx0.shape[1]) if gx1 is not None else cout
        a.gnb_ss, a.gnb_silu = _lib.ptr(gnb["ss"]), int(bool(gnb.get("silu", True)))
        if gnb.get("query_only"):
            yes = C.c_int32(0)
            _lib.check(lib.dsg_conv2d_gnb_supported(C.byref(a), C.byref(yes)))
            return bool(yes.value)
    scratch = None
    if splitk:   # small-grid calls may contract K in parallel slices (dsg_conv_args.splitk_ws)
        need = C.c_size_t()
        _lib.check(lib.dsg_conv2d_splitk_bytes(C.byref(a), C.byref(need)))
        if need.value:
            scratch = torch.empty(need.value, dtype=torch.uint8, device=src0.device)
            a.splitk_ws, a.splitk_ws_bytes = scratch.data_ptr(), need.value
    stats = None
    if want_stats and not direct and not query:
        tiles = C.c_int32(0)
        _lib.check(lib.dsg_conv2d_stats_tiles(C.byref(a), C.byref(tiles)))
        if tiles.value > 0:
            stats = stats_buf if stats_buf is not None else torch.empty(
                (n, cout, tiles.value, 2), dtype=torch.float64, device=src0.device)
            a.stats_out = stats.data_ptr()
    if isinstance(operand, str):
        yes = C.c_int32(0)
        _lib.check(lib.dsg_conv2d_takes_operand(C.byref(a), C.byref(yes)))
        if operand == "query":
            return bool(yes.value)
        operand = conv_operand_prepare(src0, src1, gn_scale_shift, silu, src_bound, src_bound1) if yes.value else None
    if operand is not None:
        a.src_operand = operand.data_ptr()
    if shortcut is not None and shortcut.get("query_only"):
        yes = C.c_int32(0)
        _lib.check(lib.dsg_conv2d_fuses_shortcut(C.byref(a), C.byref(yes)))
        return bool(yes.value)
    fn = lib.dsg_conv2d_fwd_direct if direct else lib.dsg_conv2d_fwd
    with torch.cuda.device(src0.device):
        _lib.check(fn(C.byref(a), _st(src0)))
    return (out, stats) if want_stats else out


def tuning_epoch() -> int:
    """dsg_tuning_epoch: advances with every accepted dsg_set_tuning call (a test hook) -- the key of host-side caches of
    kernel-selection answers."""
    return int(_lib.load().dsg_tuning_epoch())


def conv_operand_prepare(src0, src1=None, gn_scale_shift=None, silu=False, src_bound=None, src_bound1=None):
    """dsg_conv_operand_prepare: the pre-staged operand image [2][N][C/8][H+2][W+2][8] fp16 of cat(src0, src1)
    (channel-blocked fp32) for `conv2d_fused(..., operand=)`."""
    n, cb0, hin, win, _ = src0.shape
    c0 = 8 * cb0
    c1 = 8 * src1.shape[1] if src1 is not None else 0
    out = torch.empty((2, n, (c0 + c1) // 8, hin + 2, win + 2, 8), dtype=torch.float16, device=src0.device)
    with torch.cuda.device(src0.device):
        _lib.check(_lib.load().dsg_conv_operand_prepare(
            _lib.ptr(src0), c0, _lib.ptr(src1), c1, n, hin, win, _lib.ptr(gn_scale_shift), int(silu), _lib.ptr(src_bound),
            _lib.ptr(src_bound1), _lib.ptr(out), 0, _st(src0)))
    return out


def to_blocked(x, dtype=0):
    """fp32 [N, C, H, W] -> channel-blocked [N, C/8, H, W, 8] stored as dsg_dtype `dtype` (dsg_layout_convert_dt)."""
    n, c, h, w = x.shape
    dt = dtype_code(dtype)
    out = torch.empty((n, c // 8, h, w, 8), dtype=_lib.TORCH_DTYPES[dt], device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_layout_convert_dt(_lib.ptr(x), _lib.ptr(out), n, c, h * w, 1, dt, _st(x)))
    return out


def from_blocked(x):
    """channel-blocked [N, C/8, H, W, 8] (fp32 / bf16 / fp16) -> fp32 [N, C, H, W]."""
    n, cb, h, w, _ = x.shape
    out = torch.empty((n, cb * 8, h, w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_layout_convert_dt(_lib.ptr(x), _lib.ptr(out), n, cb * 8, h * w, 0, _DT_OF[x.dtype],
                                                     _st(x)))
    return out


def gn_channel_stats_blocked(x, splits=1):
    """dsg_gn_channel_stats_blocked_dt: per-(n, c) (sum, sum of squares) of a channel-blocked tensor (fp32 / bf16 /
    fp16) as `splits` partial sums over equal runs of pixels, fp64 [N][C][splits][2]."""
    n, cb, h, w, _ = x.shape
    st = torch.empty((n, cb * 8, splits, 2), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_gn_channel_stats_blocked_dt(_lib.ptr(x), cb * 8, n, h * w, splits, _lib.ptr(st),
                                                               _DT_OF[x.dtype], _st(x)))
    return st


def range_bound_from_stats(stats, bound=None):
    """dsg_range_bound_from_stats: per-image upper bound of max|x| (float bits in an int32 [N] tensor) from statistics
    [N][C][tiles][2] -- what dsg_conv_args.src_bound wants for a source without a norm in front."""
    n, c, tiles = stats.shape[0], stats.shape[1], stats.shape[2]
    if bound is None:
        bound = torch.zeros(n, dtype=torch.int32, device=stats.device)
    with torch.cuda.device(stats.device):
        _lib.check(_lib.load().dsg_range_bound_from_stats(_lib.ptr(stats), n, c, tiles, _lib.ptr(bound), _st(stats)))
    return bound


def gn_scale_shift_from_parts_train(stats0, gamma, beta, groups, eps, hw, stats1=None):
    """dsg_gn_finalize_parts_train: (scale_shift, mean_rstd) from per-tile partial statistics."""
    n, c0, t0 = stats0.shape[0], stats0.shape[1], stats0.shape[2]
    c1, t1 = (stats1.shape[1], stats1.shape[2]) if stats1 is not None else (0, 0)
    ss = torch.empty((n, c0 + c1, 2), dtype=torch.float32, device=stats0.device)
    mr = torch.empty((n, c0 + c1, 2), dtype=torch.float32, device=stats0.device)
    with torch.cuda.device(stats0.device):
        _lib.check(_lib.load().dsg_gn_finalize_parts_train(_lib.ptr(stats0), c0, t0, _lib.ptr(stats1), c1, t1,
                                                           _lib.ptr(gamma), _lib.ptr(beta), n, groups, hw, float(eps),
                                                           _lib.ptr(ss), _lib.ptr(mr), _st(stats0)))
    return ss, mr


def gn_scale_shift_from_parts(stats0, gamma, beta, groups, eps, hw, stats1=None):
    """dsg_gn_finalize_parts: scale/shift of GroupNorm over cat(src0, src1) from per-tile partial statistics
    [N][c_i][tiles_i][2] (conv2d_fused(want_stats=True), or channel statistics with tiles = 1)."""
    n, c0, t0 = stats0.shape[0], stats0.shape[1], stats0.shape[2]
    c1, t1 = (stats1.shape[1], stats1.shape[2]) if stats1 is not None else (0, 0)
    ss = torch.empty((n, c0 + c1, 2), dtype=torch.float32, device=stats0.device)
    with torch.cuda.device(stats0.device):
        _lib.check(_lib.load().dsg_gn_finalize_parts(_lib.ptr(stats0), c0, t0, _lib.ptr(stats1), c1, t1,
                                                     _lib.ptr(gamma), _lib.ptr(beta), n, groups, hw, float(eps),
                                                     _lib.ptr(ss), _st(stats0)))
    return ss


def gn_scale_shift(src0, gamma, beta, groups, eps, src1=None):
    """Per-(n, c) scale/shift of GroupNorm over cat(src0, src1): [N][C][2]."""
    lib = _lib.load()
    n, c0 = src0.shape[0], src0.shape[1]
    hw = src0.shape[2] * src0.shape[3] if src0.dim() == 4 else src0.shape[2]
    c1 = src1.shape[1] if src1 is not None else 0
    c = c0 + c1
    stats = torch.empty((n, c, 2), dtype=torch.float64, device=src0.device)
    ss = torch.empty((n, c, 2), dtype=torch.float32, device=src0.device)
    with torch.cuda.device(src0.device):
        _lib.check(lib.dsg_gn_channel_stats(_lib.ptr(src0), c0, _lib.ptr(src1), c1, n, hw, _lib.ptr(stats),
                                            _st(src0)))
        _lib.check(lib.dsg_gn_finalize(_lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta), n, c, groups, hw,
                                       float(eps), _lib.ptr(ss), _st(src0)))
    return ss


def gn_apply(src, scale_shift, silu=False):
    out = torch.empty_like(src)
    n, c = src.shape[0], src.shape[1]
    hw = src.numel() // (n * c)
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().dsg_gn_apply(_lib.ptr(src), _lib.ptr(scale_shift), int(silu), _lib.ptr(out), n, c,
                                            hw, _st(src)))
    return out


def attention(qkv, heads, dtype=0):
    """qkv [N, 3C, L] -> [N, C, L]; dtype: dsg_dtype of the matrix-core products (fp32 tensors in every mode)."""
    n, c3, l = qkv.shape
    c = c3 // 3
    out = torch.empty((n, c, l), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        _lib.check(_lib.load().dsg_attention_fwd_dt(_lib.ptr(qkv), _lib.ptr(out), n, c, heads, l, dtype_code(dtype), _st(qkv)))
    return out


def attention_blocked(qkv_blk, heads, dtype=0):
    """dsg_attention_fwd_blocked: channel-blocked qkv [N, 3C/8, H, W, 8] (fp32, or the 16-bit type of `dtype`) -> blocked
    [N, C/8, H, W, 8] of the same element type.  head_dim 8 (one channel block per head), H * W % 32 == 0."""
    n, cb3, h, w, _ = qkv_blk.shape
    c, l = cb3 * 8 // 3, h * w
    out = torch.empty((n, c // 8, h, w, 8), dtype=qkv_blk.dtype, device=qkv_blk.device)
    with torch.cuda.device(qkv_blk.device):
        _lib.check(_lib.load().dsg_attention_fwd_blocked(_lib.ptr(qkv_blk), _lib.ptr(out), n, c, heads, l,
                                                        dtype_code(dtype), _st(qkv_blk)))
    return out


def sinusoid_freqs(ch: int) -> torch.Tensor:
    """exp(-ln(10000) * arange(ch/2) / (ch/2)) with the reference's fp32 op order (get_timestep_embedding,
    flip_sin_to_cos=True, freq_shift=0).  Host-side, [ch/2] fp32 CPU tensor."""
    import math
    half = ch // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32)
    return torch.exp(exponent / half)


def time_embed(timesteps, w1, b1, w2, b2, freqs=None):
    """silu(linear_2(silu(linear_1(sinusoid(t))))) -> [N, dim]."""
    n = timesteps.numel()
    dim, ch = w1.shape
    if freqs is None:
        freqs = sinusoid_freqs(ch).to(w1.device)
    act = torch.empty((n, dim), dtype=torch.float32, device=w1.device)
    with torch.cuda.device(w1.device):
        _lib.check(_lib.load().dsg_time_embed_fwd(_lib.ptr(timesteps), _lib.ptr(freqs), n, ch, dim, _lib.ptr(w1), _lib.ptr(b1),
                                                 _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(act), _st(w1)))
    return act


def linear(x, w, b=None):
    n, in_f = x.shape
    out_f = w.shape[0]
    y = torch.empty((n, out_f), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_linear_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), n, in_f, out_f,
                                             _st(x)))
    return y


def postprocess(x, mode=0):
    """(x/2+0.5).clamp(0,1), NCHW->NHWC; mode 0 float, 1 uint8 round, 2 uint8 truncation."""
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, c), dtype=torch.float32 if mode == 0 else torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_postprocess(_lib.ptr(x.contiguous()), out.data_ptr(), b, c, h * w, mode, _st(x)))
    return out


# ---------------------------------------------------------------------------------------------------
# Training-step ops (backward / loss / optimizer); see include/dsg.h "Training step"
# ---------------------------------------------------------------------------------------------------
def philox_u32(numel: int, seed: int, offset: int, device="cuda") -> torch.Tensor:
    """The first `numel` uint32 (as an int32-typed tensor of raw bits) of the counter-based stream (seed, offset):
    ``dsg_philox_u32`` -- Philox4x32-10, the generator behind ``DDPMScheduler.add_noise_device``."""
    out = torch.empty(numel, dtype=torch.int32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().dsg_philox_u32(_lib.ptr(out), numel, int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1),
                                             _lib.stream_ptr(out.device)))
    return out


def philox_normal(shape, seed: int, offset: int, device="cuda") -> torch.Tensor:
    """fp32 N(0, 1) tensor of `shape` from the counter-based stream (seed, offset) (``dsg_philox_normal``)."""
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().dsg_philox_normal(_lib.ptr(out), out.numel(), int(seed) & (2 ** 64 - 1),
                                                int(offset) & (2 ** 64 - 1), _lib.stream_ptr(out.device)))
    return out


def wgrad16_supported(c0, c1, cout, h, w, ksize=3, stride=1, upsample=False, dy_coff=0) -> bool:
    """Shapes dsg_conv2d_wgrad serves on channel-blocked 16-bit tensors (include/dsg.h); the rest goes through
    from_blocked() and the fp32 form."""
    chans = (stride in (1, 2) and not (upsample and stride == 2) and (c0 + c1) % 64 == 0 and (c1 == 0 or c0 % 64 == 0)
             and cout % 64 == 0 and dy_coff % 64 == 0)
    if upsample or stride == 2:
        if os.environ.get("DSG_W16_SAMPLER") == "0":   # A/B hook (tools/collect_r06.sh): the round-5 route on materialised tensors
            return False
        # the sampler convs: h, w = the source map; one operand at half the K grid's resolution (x behind Upsample2D, dY of the
        # stride-2 conv), addressed through a shift inside the kernel -- no materialised x2 copy, no zero-stuffed dY
        kh, kw = (2 * h, 2 * w) if upsample else (h, w)
        return chans and ksize == 3 and c1 == 0 and kw % 64 == 0 and kh % 4 == 0
    if ksize == 1:
        return chans and (h * w) % 64 == 0
    return chans and ksize == 3 and w % 32 == 0 and h % 2 == 0


def wgrad_h2_supported(c0, c1, cout, h, w, ksize=3, stride=1, upsample=False) -> bool:
    """fp32 [N, C, H, W] calls of dsg_conv2d_wgrad that the fp16x2-split 3x3 kernel serves (the ones that can return
    dy_sums as a by-product); h, w: the conv's output map."""
    return (ksize == 3 and stride == 1 and not upsample and (c0 + c1) % 32 == 0 and cout % 64 == 0 and w % 32 == 0 and h % 2 == 0
            and (c1 == 0 or c0 % 32 == 0))


def conv_wgrad(src0, dy, dw, src1=None, ksize=3, stride=1, upsample=False, gn_scale_shift=None, silu=False,
               direct=False, cout=None, dy_coff=0, dy_sums=None, dy_sums_stride=0, bias_grad=None):
    """dw[cout][cin][k][k] += wgrad; the activation is recomputed from (src, gn_scale_shift).  Channel-blocked 16-bit
    src / dy tensors ([N, C/8, H, W, 8] bf16 / fp16, the mixed-precision tape) select the 16-bit kernel."""
    a = _lib.ConvWgradArgs()
    if src0.dim() == 5:
        n, cb0, hin, win, _ = src0.shape
        c0, c1 = 8 * cb0, (8 * src1.shape[1] if src1 is not None else 0)
        dy_c = 8 * dy.shape[1]
        a.compute_dtype = _DT_OF[src0.dtype]
        if dy.dtype != src0.dtype or dy.dim() != 5:
            raise RuntimeError("conv_wgrad: src and dy must both be channel-blocked tensors of one 16-bit dtype")
    else:
        n, c0, hin, win = src0.shape
        c1 = src1.shape[1] if src1 is not None else 0
        dy_c = dy.shape[1]
    a.src0, a.src1 = _lib.ptr(src0), _lib.ptr(src1)
    a.c0, a.c1 = c0, c1
    a.n, a.hin, a.win = n, hin, win
    a.upsample, a.ksize, a.stride, a.cout = int(upsample), ksize, stride, cout or dy_c
    a.dy, a.gn_scale_shift, a.silu = _lib.ptr(dy), _lib.ptr(gn_scale_shift), int(silu)
    a.dy_ctotal, a.dy_coff = dy_c, dy_coff
    a.dw, a.force_direct = _lib.ptr(dw), int(direct)
    if dy_sums is not None:   # per-(n, cout) sums of dy as a by-product (bias / temb gradients): see wgrad*_supported
        a.dy_sums, a.dy_sums_stride = dy_sums.data_ptr(), int(dy_sums_stride or dy_sums.stride(0))
        if bias_grad is not None:   # ... and their sum over the batch straight into the bias gradient (dy_bias_grad)
            a.dy_bias_grad = bias_grad.data_ptr()
    lib = _lib.load()
    need = C.c_size_t()
    _lib.check(lib.dsg_conv2d_wgrad_workspace_bytes(C.byref(a), C.byref(need)))
    ws = _wgrad_ws(src0.device, need.value)
    a.workspace, a.workspace_bytes = (ws.data_ptr(), ws.numel()) if ws is not None else (None, 0)
    with torch.cuda.device(src0.device):
        _lib.check(lib.dsg_conv2d_wgrad(C.byref(a), _st(src0)))
    return dw


_WGRAD_WS = {}


def _wgrad_ws(device, nbytes):
    """Cached split-K workspace (grows to the largest layer; stream order makes reuse across calls safe)."""
    if nbytes == 0:
        return None
    key = str(device)
    cur = _WGRAD_WS.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _WGRAD_WS[key] = cur
    return cur


def gn_scale_shift_train(src0, gamma, beta, groups, eps, src1=None):
    """(scale_shift [N][C][2], mean_rstd [N][C][2]) of GroupNorm over cat(src0, src1)."""
    lib = _lib.load()
    n, c0 = src0.shape[0], src0.shape[1]
    hw = src0.numel() // (n * c0)
    c1 = src1.shape[1] if src1 is not None else 0
    c = c0 + c1
    stats = torch.empty((n, c, 2), dtype=torch.float64, device=src0.device)
    ss = torch.empty((n, c, 2), dtype=torch.float32, device=src0.device)
    mr = torch.empty((n, c, 2), dtype=torch.float32, device=src0.device)
    with torch.cuda.device(src0.device):
        _lib.check(lib.dsg_gn_channel_stats(_lib.ptr(src0), c0, _lib.ptr(src1), c1, n, hw, _lib.ptr(stats),
                                            _st(src0)))
        _lib.check(lib.dsg_gn_finalize_train(_lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta), n, c, groups, hw,
                                             float(eps), _lib.ptr(ss), _lib.ptr(mr), _st(src0)))
    return ss, mr


def gn_bwd(src0, dy, ss, mr, gamma, groups, silu, dgamma, dbeta, src1=None, add0=None, add1=None, parts=None, add0b=None):
    """Backward of silu?(GroupNorm(cat(src0, src1))); returns (dx0, dx1); dgamma/dbeta accumulated.
    parts: the [N][C][tiles][2] table a data-gradient conv's GNB epilogue wrote (conv2d_fused(gnb=...)): replaces the
    statistics pass over x and dy."""
    n, c0 = src0.shape[0], src0.shape[1]
    hw = src0.numel() // (n * c0)
    c1 = src1.shape[1] if src1 is not None else 0
    c = c0 + c1
    dx0 = torch.empty_like(src0)
    dx1 = torch.empty_like(src1) if src1 is not None else None
    s12 = torch.empty((n, c, 2), dtype=torch.float64, device=src0.device)
    coef = torch.empty((n, c, 3), dtype=torch.float32, device=src0.device)
    with torch.cuda.device(src0.device):
        if add0b is not None:    # a second waiting gradient of source 0, added in the same pass
            _lib.check(_lib.load().dsg_gn_bwd_add2(_lib.ptr(src0), c0, _lib.ptr(src1), c1, _lib.ptr(dy), _lib.ptr(ss),
                                                  _lib.ptr(mr), _lib.ptr(gamma), int(silu), n, hw, groups, _lib.ptr(add0),
                                                  _lib.ptr(add0b), _lib.ptr(add1), _lib.ptr(dx0), _lib.ptr(dx1), _lib.ptr(dgamma),
                                                  _lib.ptr(dbeta), _lib.ptr(s12), _lib.ptr(coef), _lib.ptr(parts),
                                                  parts.shape[2] if parts is not None else 0, _st(src0)))
        elif parts is not None:
            _lib.check(_lib.load().dsg_gn_bwd_parts(_lib.ptr(src0), c0, _lib.ptr(src1), c1, _lib.ptr(dy), _lib.ptr(ss),
                                                   _lib.ptr(mr), _lib.ptr(gamma), int(silu), n, hw, groups, _lib.ptr(add0),
                                                   _lib.ptr(add1), _lib.ptr(dx0), _lib.ptr(dx1), _lib.ptr(dgamma),
                                                   _lib.ptr(dbeta), _lib.ptr(s12), _lib.ptr(coef), _lib.ptr(parts),
                                                   parts.shape[2], _st(src0)))
        else:
            _lib.check(_lib.load().dsg_gn_bwd(_lib.ptr(src0), c0, _lib.ptr(src1), c1, _lib.ptr(dy), _lib.ptr(ss),
                                             _lib.ptr(mr), _lib.ptr(gamma), int(silu), n, hw, groups, _lib.ptr(add0),
                                             _lib.ptr(add1), _lib.ptr(dx0), _lib.ptr(dx1), _lib.ptr(dgamma),
                                             _lib.ptr(dbeta), _lib.ptr(s12), _lib.ptr(coef), _st(src0)))
    return dx0, dx1


def gn_bwd_blocked(src0, dy, ss, mr, gamma, groups, silu, dgamma, dbeta, src1=None, add0=None, add1=None, add0b=None,
                   parts=None):
    """gn_bwd on channel-blocked 16-bit tensors [N, C/8, H, W, 8] (dy covers cat(src0, src1)); returns (dx0, dx1).
    add0 / add0b / add1: gradients already waiting on the sources (fan-in), added in the same pass.
    parts: as in gn_bwd."""
    n, cb0, h, w, _ = src0.shape
    c0, c1 = 8 * cb0, (8 * src1.shape[1] if src1 is not None else 0)
    c, hw = c0 + c1, h * w
    lib = _lib.load()
    splits = lib.dsg_gn_bwd_blocked_splits(hw)
    dx0 = torch.empty_like(src0)
    dx1 = torch.empty_like(src1) if src1 is not None else None
    s12 = torch.empty(n * c * 2 * (1 + splits), dtype=torch.float64, device=src0.device)
    coef = torch.empty((n, c, 3), dtype=torch.float32, device=src0.device)
    if parts is not None:
        s12 = torch.empty(n * c * 2, dtype=torch.float64, device=src0.device)
        with torch.cuda.device(src0.device):
            _lib.check(lib.dsg_gn_bwd_blocked_parts(_lib.ptr(src0), c0, _lib.ptr(src1), c1, _lib.ptr(dy), _lib.ptr(ss), _lib.ptr(mr),
                                                    _lib.ptr(gamma), int(silu), n, hw, groups, _lib.ptr(add0), _lib.ptr(add0b),
                                                    _lib.ptr(add1), _lib.ptr(dx0), _lib.ptr(dx1), _lib.ptr(dgamma), _lib.ptr(dbeta),
                                                    _lib.ptr(s12), _lib.ptr(coef), _DT_OF[src0.dtype], _lib.ptr(parts),
                                                    parts.shape[2], _st(src0)))
        return dx0, dx1
    with torch.cuda.device(src0.device):
        _lib.check(lib.dsg_gn_bwd_blocked_add2(_lib.ptr(src0), c0, _lib.ptr(src1), c1, _lib.ptr(dy), _lib.ptr(ss), _lib.ptr(mr),
                                               _lib.ptr(gamma), int(silu), n, hw, groups, _lib.ptr(add0), _lib.ptr(add0b),
                                               _lib.ptr(add1), _lib.ptr(dx0), _lib.ptr(dx1), _lib.ptr(dgamma), _lib.ptr(dbeta),
                                               _lib.ptr(s12), _lib.ptr(coef), _DT_OF[src0.dtype], _st(src0)))
    return dx0, dx1


def channel_sums(x, out=None, out_stride=None):
    """[N, C, ...] -> [N, C] sums over the trailing dims (optionally into rows of a wider matrix); channel-blocked
    16-bit tensors [N, C/8, H, W, 8] are summed per channel likewise."""
    if x.dim() == 5:
        n, cb, h, w, _ = x.shape
        if out is None:
            out = torch.empty((n, cb * 8), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dsg_channel_sums_blocked(_lib.ptr(x), n, cb * 8, h * w, out.data_ptr(),
                                                            out_stride or out.stride(0), _DT_OF[x.dtype], _st(x)))
        return out
    n, c = x.shape[0], x.shape[1]
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_channel_sums(_lib.ptr(x), n, c, x.numel() // (n * c), out.data_ptr(),
                                               out_stride or out.stride(0), _st(x)))
    return out


def upsample_nearest2x(x):
    """[N, C, h, w] -> [N, C, 2h, 2w] (dsg_upsample_nearest2x); channel-blocked 16-bit [N, C/8, h, w, 8] likewise."""
    if x.dim() == 5:
        n, cb, h, w, _ = x.shape
        out = torch.empty((n, cb, 2 * h, 2 * w, 8), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dsg_upsample_nearest2x_blocked(_lib.ptr(x), _lib.ptr(out), n * cb, h, w, _DT_OF[x.dtype], _st(x)))
        return out
    n, c, h, w = x.shape
    out = torch.empty((n, c, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_upsample_nearest2x(_lib.ptr(x), _lib.ptr(out), n * c, h, w, _st(x)))
    return out


def sumpool2x2(x, add=None):
    """[N, C, 2h, 2w] -> [N, C, h, w] sums of 2x2 blocks (+ add): the adjoint of upsample_nearest2x."""
    if x.dim() == 5:
        n, cb, h2, w2, _ = x.shape
        out = torch.empty((n, cb, h2 // 2, w2 // 2, 8), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dsg_sumpool2x2_blocked(_lib.ptr(x), _lib.ptr(add), _lib.ptr(out), n * cb, h2 // 2, w2 // 2,
                                                          _DT_OF[x.dtype], _st(x)))
        return out
    n, c, h2, w2 = x.shape
    out = torch.empty((n, c, h2 // 2, w2 // 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_sumpool2x2(_lib.ptr(x), _lib.ptr(add), _lib.ptr(out), n * c, h2 // 2, w2 // 2, _st(x)))
    return out


def add(a, b):
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().dsg_add_dt(_lib.ptr(a), _lib.ptr(b), a.numel(), _lib.ptr(out), _DT_OF[a.dtype], _st(a)))
    return out


def time_embed_train(timesteps, w1, b1, w2, b2, freqs):
    """(act, emb, z1, z2) -- act = silu(z2), z2 = linear_2(silu(z1)), z1 = linear_1(emb)."""
    n = timesteps.numel()
    dim, ch = w1.shape
    dev = w1.device
    act, z1, z2 = (torch.empty((n, dim), dtype=torch.float32, device=dev) for _ in range(3))
    emb = torch.empty((n, ch), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().dsg_time_embed_fwd_train(_lib.ptr(timesteps), _lib.ptr(freqs), n, ch, dim, _lib.ptr(w1),
                                                       _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(act),
                                                       _lib.ptr(emb), _lib.ptr(z1), _lib.ptr(z2), _st(w1)))
    return act, emb, z1, z2


def reduce_rows_add(src, dst, stride=None):
    """dst[c] += sum_n src[n, c]  (src rows `stride` floats apart)."""
    n = src.shape[0]
    c = dst.numel()
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().dsg_reduce_rows_add(src.data_ptr(), n, c, stride or src.stride(0), _lib.ptr(dst),
                                                   _st(src)))
    return dst


def attention_train(qkv, heads, dtype=0):
    """attention core + the log2-domain log-sum-exp the backward recomputes from; dtype "bf16": the mixed-precision tape's
    arithmetic (operands rounded once to bf16 on the matrix cores)."""
    n, c3, l = qkv.shape
    c = c3 // 3
    out = torch.empty((n, c, l), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((n, heads, l), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        _lib.check(_lib.load().dsg_attention_fwd_train_dt(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(lse), n, c, heads, l,
                                                         dtype_code(dtype), _st(qkv)))
    return out, lse


def attention_bwd(qkv, out, dout, lse, heads, dtype=0):
    """dsg_attention_bwd_dt: gradient of the attention core w.r.t. the fused [N, 3C, L] q / k / v tensor.  dtype "bf16": the
    mixed-precision tape's arithmetic (matrix cores for head_dim 8; everything else and dtype 0: the exact kernels)."""
    n, c3, l = qkv.shape
    c = c3 // 3
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty((n, heads, l), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        _lib.check(_lib.load().dsg_attention_bwd_dt(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(dout.contiguous()),
                                                   _lib.ptr(lse), _lib.ptr(dqkv), _lib.ptr(dsum), n, c, heads, l,
                                                   dtype_code(dtype), _st(qkv)))
    return dqkv


def linear_bwd(x, w, dy, dw=None, db=None, need_dx=True, dy_stride=None):
    """Backward of y = x W^T + b; dw/db accumulated in place; returns dx or None."""
    n, in_f = x.shape
    out_f = w.shape[0]
    dx = torch.empty((n, in_f), dtype=torch.float32, device=x.device) if need_dx else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_linear_bwd(_lib.ptr(x), _lib.ptr(w), dy.data_ptr(), dy_stride or dy.stride(0), n,
                                             in_f, out_f, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(dx), _st(x)))
    return dx


def silu_fwd(z):
    y = torch.empty_like(z)
    with torch.cuda.device(z.device):
        _lib.check(_lib.load().dsg_silu_fwd(_lib.ptr(z), z.numel(), _lib.ptr(y), _st(z)))
    return y


def silu_bwd(z, dy):
    dz = torch.empty_like(z)
    with torch.cuda.device(z.device):
        _lib.check(_lib.load().dsg_silu_bwd(_lib.ptr(z), _lib.ptr(dy.contiguous()), z.numel(), _lib.ptr(dz), _st(z)))
    return dz


_WS = {}


def _reduce_ws(device):
    key = str(device)
    if key not in _WS:
        _WS[key] = torch.empty(2048, dtype=torch.float64, device=device)
    return _WS[key]


def mse_loss(pred, target, grad_scale=1.0, need_grad=True):
    """(loss [1] fp32 device tensor, dpred or None): F.mse_loss(pred, target) and its gradient * grad_scale."""
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if need_grad else None
    ws = _reduce_ws(pred.device)
    with torch.cuda.device(pred.device):
        _lib.check(_lib.load().dsg_mse_loss(_lib.ptr(pred.contiguous()), _lib.ptr(target.contiguous()), pred.numel(),
                                           float(grad_scale), _lib.ptr(loss), _lib.ptr(dpred), _lib.ptr(ws),
                                           ws.numel() * 8, _st(pred)))
    return loss, dpred


def l2_norm(x):
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = _reduce_ws(x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_l2_norm(_lib.ptr(x), x.numel(), _lib.ptr(out), _lib.ptr(ws), ws.numel() * 8,
                                          _st(x)))
    return out


def unscale_check_(g, inv_scale, found_inf):
    """GradScaler.unscale_: g *= inv_scale in place; found_inf (device int32 [1]) |= any non-finite element."""
    with torch.cuda.device(g.device):
        _lib.check(_lib.load().dsg_unscale_check(_lib.ptr(g), g.numel(), float(inv_scale), found_inf.data_ptr(), _st(g)))
    return g


def clip_scale_(g, total_norm, max_norm):
    with torch.cuda.device(g.device):
        _lib.check(_lib.load().dsg_clip_scale(_lib.ptr(g), g.numel(), _lib.ptr(total_norm), float(max_norm), _st(g)))
    return g


def adamw_step_(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                total_norm=None, max_norm=0.0):
    with torch.cuda.device(param.device):
        _lib.check(_lib.load().dsg_adamw_step(_lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg),
                                             _lib.ptr(exp_avg_sq), param.numel(), float(lr), float(betas[0]),
                                             float(betas[1]), float(eps), float(weight_decay), int(step),
                                             _lib.ptr(total_norm), float(max_norm), _st(param)))


def scale(x, alpha_dev=None, mult=1.0, out=None):
    """out = x * alpha_dev[0] * mult (in place when out is x)."""
    out = torch.empty_like(x) if out is None else out
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_scale(_lib.ptr(x), x.numel(), _lib.ptr(alpha_dev), float(mult), _lib.ptr(out),
                                        _st(x)))
    return out
