"""GroupNorm-backward statistics from the data-gradient conv's epilogue (dsg_conv_args.gnb_*, conv_h2_kernel's GNB form).

Reference path: `accelerator.backward(loss)` (training_pipeline.py:86) through ResnetBlock2D's norm1 / norm2: the gradient w.r.t.
the activated tensor a = silu(GroupNorm(x)) comes out of the conv's data gradient, and the norm's backward needs per-(n, c) sums of
du = dA * silu'(x * sc + sh) and du * xhat -- a pass of its own over x and dA (gn_bwd_stats*_kernel).  With gnb_* the conv's
epilogue leaves per-tile (sum du, sum du * x) and ``dsg_gn_bwd*_parts`` finishes from them.  Checked here: dA is bitwise the
plain call's; dgamma / dbeta / dx equal the statistics-pass path to summation-order round-off and an fp64 autograd evaluation in
the same class; every kernel geometry that has the form; the shapes that do not say so."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from drivescenegen_amd import _lib, ops, synth  # noqa: E402

DEV = "cuda"
GROUPS = 32


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _case(n, cdy, c0, c1, h, w, silu, seed):
    """dY [n, cdy, h, w] of a conv whose input was silu?(GroupNorm(cat(x0, x1))), x = c0 + c1 channels with a mean far from 0
    (the raw second moment's cancellation is exercised), W [cdy, c0 + c1, 3, 3], gamma / beta."""
    c = c0 + c1
    x = _t(seed, (n, c, h, w), 1.3) + 0.7 * _t(seed + 1, (1, c, 1, 1)) + 1.5
    dy = _t(seed + 2, (n, cdy, h, w), 0.05)
    wt = _t(seed + 3, (cdy, c, 3, 3), 1.0 / np.sqrt(9 * c))
    gamma, beta = 1 + _t(seed + 4, (c,), 0.2), _t(seed + 5, (c,), 0.2)
    return x, dy, wt, gamma, beta


def _ref64(x, dy, wt, gamma, beta, silu):
    """fp64 autograd: dx, dgamma, dbeta of L = <conv(act(gn(x))), dy>."""
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.group_norm(x64, GROUPS, g64, b64, 1e-5)
    if silu:
        a = F.silu(a)
    y = F.conv2d(a, wt.double(), None, padding=1)
    (y * dy.double()).sum().backward()
    return x64.grad, g64.grad, b64.grad


# (name, n, cdy, c0, c1, h, w, silu, dtype): which kernel geometry serves the data gradient (cout of that call = c0 + c1)
CASES = [
    ("bf16_bm128_r16", 8, 64, 128, 0, 64, 64, True, "bf16"),        # 128-cout workgroups, 16-row tiles
    ("bf16_bm128_r8", 8, 128, 256, 0, 64, 64, True, "bf16"),        # 128-cout workgroups, 8-row tiles (256 of them; 128 of 16 rows)
    ("bf16_bm64_nt2_c32", 16, 32, 32, 0, 64, 64, True, "bf16"),     # configs[0]'s shapes: 32 channels = half a 64-cout tile, 8-row tiles
    ("bf16_bm64_nt4", 8, 64, 64, 0, 128, 128, True, "bf16"),        # 64-cout workgroups (cout % 128 != 0), 16-row tiles
    ("bf16_cat_128_64", 8, 64, 128, 64, 64, 64, True, "bf16"),      # concatenated x: tiles of 64 ... 192 % 128 != 0 -> BM 64, c0 % 64 == 0
    ("bf16_cat_256_128", 4, 128, 256, 128, 64, 64, True, "bf16"),   # BM 128, c0 % 128 == 0
    ("fp16_bm128", 8, 64, 128, 0, 64, 64, True, "fp16"),
    ("bf16_affine_only", 8, 64, 128, 0, 64, 64, False, "bf16"),     # GroupNorm without SiLU (gnb_silu = 0)
    ("fp32_nt4", 4, 64, 128, 0, 64, 64, True, "fp32"),              # the fp32 tape: [N, C, H, W] tensors, 16-row tiles
    ("fp32_cat", 2, 128, 128, 64, 64, 64, True, "fp32"),
    ("fp32_nt2", 1, 64, 64, 0, 32, 32, True, "fp32"),               # small grid: 8-row tiles
]


@pytest.fixture(params=[1, 0], ids=["gnb_on_64_cout_workgroups", "gnb_on_128_cout_workgroups"])
def gnb_workgroups(request):
    """dsg_set_tuning key 41: the 16-bit GNB data gradients run on 64-cout workgroups by default; 0 lets them take the 128-cout
    ones the case names describe (both instantiations stay covered)"""
    lib = _lib.load()
    _lib.check(lib.dsg_set_tuning(41, request.param))
    yield request.param
    lib.dsg_set_tuning(41, 1)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_epilogue_statistics_equal_the_statistics_pass(case, gnb_workgroups):
    name, n, cdy, c0, c1, h, w, silu, dtn = case
    x, dy, wt, gamma, beta = _case(n, cdy, c0, c1, h, w, silu, seed=sum(map(ord, name)))
    c = c0 + c1
    g = lambda t: t.to(DEV)
    dt = ops.dtype_code(dtn)
    blocked = dtn != "fp32"
    tol_dx = 1e-5 if not blocked else (3e-3 if dtn == "bf16" else 5e-4)
    if blocked:
        xb = ops.to_blocked(g(x), dt)
        x0b = xb[:, :c0 // 8].contiguous()
        x1b = xb[:, c0 // 8:].contiguous() if c1 else None
        dyb = ops.to_blocked(g(dy), dt)
        xr = ops.from_blocked(xb).cpu()          # the 16-bit x the kernels see
        dyr = ops.from_blocked(dyb).cpu()
    else:
        x0b, x1b, dyb = g(x[:, :c0]).contiguous(), (g(x[:, c0:]).contiguous() if c1 else None), g(dy)
        xr, dyr = x, dy
    ss, mr = ops.gn_scale_shift_train(g(xr[:, :c0]).contiguous(), g(gamma), g(beta), GROUPS, 1e-5,
                                      src1=g(xr[:, c0:]).contiguous() if c1 else None)
    wd = ops.relayout_conv_weight_dgrad(g(wt))
    whd = ops.pack_conv_weight(g(wt), ops.PACK_DGRAD, dt) if blocked else ops.relayout_conv_weight_h2_dgrad(g(wt))
    kw = dict(ksize=3, cout=c, src_blocked=blocked, dst_blocked=blocked, compute_dtype=dt, weight_h2=whd)
    if blocked:
        kw["weight_h2_stride"] = (c + 63) // 64 * 64
    gnb = dict(x0=x0b, x1=x1b, ss=ss, silu=silu)
    assert ops.conv2d_fused(dyb, wd, gnb=dict(gnb, query_only=True), **kw), "this shape should take the GNB kernel"
    da_plain = ops.conv2d_fused(dyb, wd, **kw)
    da, parts = ops.conv2d_fused(dyb, wd, gnb=gnb, want_stats=True, **kw)
    assert parts is not None and parts.shape[:2] == (n, c) and parts.shape[3] == 2
    assert torch.equal(da, da_plain)                          # dst is untouched by the statistics
    bwd = ops.gn_bwd_blocked if blocked else ops.gn_bwd
    res = {}
    for key, p in (("pass", None), ("parts", parts)):
        dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        dx0, dx1 = bwd(x0b, da, ss, mr, g(gamma), GROUPS, silu, dg, db, src1=x1b, parts=p)
        dxf = torch.cat([ops.from_blocked(t) if blocked else t for t in (dx0, dx1) if t is not None], 1).cpu()
        res[key] = (dxf, dg.cpu(), db.cpu())
    # the two paths: summation order + the raw-moment transform (fp64 on fp32 tile partials) -- far inside either's distance
    # to the exact value; dx is rounded to the tape's 16-bit type (a 1-ulp flip here and there)
    assert _rel(res["parts"][1], res["pass"][1]) <= 2e-5, _rel(res["parts"][1], res["pass"][1])
    assert _rel(res["parts"][2], res["pass"][2]) <= 2e-5
    assert _rel(res["parts"][0], res["pass"][0]) <= tol_dx
    # fp64 autograd on the operands the kernels saw (x and dA as stored)
    daf = (ops.from_blocked(da) if blocked else da).cpu()
    x64 = xr.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.group_norm(x64, GROUPS, g64, b64, 1e-5)
    if silu:
        a = F.silu(a)
    (a * daf.double()).sum().backward()
    for key in ("pass", "parts"):
        assert _rel(res[key][0], x64.grad) <= max(tol_dx, 2e-5), (key, _rel(res[key][0], x64.grad))
        assert _rel(res[key][1], g64.grad) <= 5e-5, (key, _rel(res[key][1], g64.grad))
        assert _rel(res[key][2], b64.grad) <= 5e-5, (key, _rel(res[key][2], b64.grad))
    err_parts, err_pass = _rel(res["parts"][1], g64.grad), _rel(res["pass"][1], g64.grad)
    assert err_parts <= 4 * err_pass + 2e-6                   # the epilogue's sums are in the statistics pass's rounding class


def test_shapes_without_the_form_say_so_and_the_switch_turns_it_off():
    g = lambda t: t.to(DEV)
    dt = ops.dtype_code("bf16")

    def ask(n, cdy, c0, c1, h, w, **extra):
        x, dy, wt, gamma, beta = _case(n, cdy, c0, c1, h, w, True, 5)
        c = c0 + c1
        xb = ops.to_blocked(g(x), dt)
        x0b = xb[:, :c0 // 8].contiguous()
        x1b = xb[:, c0 // 8:].contiguous() if c1 else None
        ss, _ = ops.gn_scale_shift_train(g(x[:, :c0]).contiguous(), g(gamma), g(beta), GROUPS, 1e-5,
                                         src1=g(x[:, c0:]).contiguous() if c1 else None)
        kw = dict(ksize=3, cout=c, src_blocked=True, dst_blocked=True, compute_dtype=dt,
                  weight_h2=ops.pack_conv_weight(g(wt), ops.PACK_DGRAD, dt), weight_h2_stride=(c + 63) // 64 * 64)
        kw.update(extra)
        gnb = dict(x0=x0b, x1=x1b, ss=ss, silu=True)
        dyb = ops.to_blocked(g(dy), dt)
        return ops.conv2d_fused(dyb, None, gnb=dict(gnb, query_only=True), **kw), (dyb, gnb, kw)
    assert ask(8, 64, 128, 0, 64, 64)[0]
    assert not ask(1, 64, 64, 0, 32, 32)[0]            # a grid of at most half the chip: the 32-cout workgroups have no GNB form
    # cat(192, 64) / configs[0]'s cat(64, 32) / cat(96, 32): the x tensors do not meet at a multiple of 128 channels -- such calls
    # run on 64-cout workgroups, and where the seam falls inside one of those (96 | 32) the epilogue reads x slab by slab (32
    # channels) from the tensor that holds it (values: test_query_and_dispatch_agree_over_a_shape_sweep); key 37 = 3 refuses
    # them, as round 6's first rule did
    assert ask(8, 64, 192, 64, 64, 64)[0] and ask(16, 32, 64, 32, 64, 64)[0] and ask(8, 64, 96, 32, 64, 64)[0]
    lib = _lib.load()
    try:
        _lib.check(lib.dsg_set_tuning(37, 3))
        assert not ask(8, 64, 96, 32, 64, 64)[0]           # the seam inside a 64-cout tile
        assert ask(8, 64, 192, 64, 64, 64)[0]              # ... between two 64-cout tiles (key 41: the GNB calls' workgroups)
        _lib.check(lib.dsg_set_tuning(41, 0))              # ... inside a 128-cout one
        yes, (dyb, gnb, kw) = ask(8, 64, 192, 64, 64, 64)
        assert not yes
        with pytest.raises(RuntimeError, match="GroupNorm-backward epilogue"):
            ops.conv2d_fused(dyb, None, gnb=gnb, want_stats=True, **kw)
        assert ask(8, 64, 128, 0, 64, 64)[0]
    finally:
        lib.dsg_set_tuning(37, 1)
        lib.dsg_set_tuning(41, 1)
    assert not ask(8, 64, 112, 16, 64, 64)[0]          # cat(112, 16): the seam is inside a 32-channel slab
    assert not ask(8, 64, 128, 0, 64, 48)[0]           # not a multiple of 32 columns
    try:
        _lib.check(lib.dsg_set_tuning(37, 0))
        assert not ask(8, 64, 128, 0, 64, 64)[0]
    finally:
        lib.dsg_set_tuning(37, 1)


@pytest.mark.parametrize("dtn", ["fp32", "bf16"])
def test_training_step_gradients_with_and_without_the_epilogue_statistics(dtn):
    """The whole backward walk on configs[0]'s network at a batch large enough for the GNB kernels (tuning key 37 = 1, default)
    against key 37 = 0 (the statistics pass): the same loss bit for bit (the forward is untouched), every gradient equal to
    summation-order round-off."""
    import drivescenegen_amd as d
    from tests.common import CFG1, synth_weights
    lib = _lib.load()
    grads = {}
    x0 = torch.from_numpy(synth.synth_scene_rasters(16, 3, 64, 64, 3)).to(DEV)
    nz = torch.from_numpy(synth.normal(4, (16, 3, 64, 64))).to(DEV)
    t = torch.arange(16, device=DEV) * 60
    sch = d.DDPMScheduler()
    try:
        for on in (1, 0):
            _lib.check(lib.dsg_set_tuning(37, on))
            net = synth_weights(d.UNet2DModel(**CFG1)).to(DEV).train().set_compute_dtype(dtn)
            loss = d.mse_loss(net(sch.add_noise(x0, nz, t), t, return_dict=False)[0], nz)
            loss.backward()
            grads[on] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    finally:
        lib.dsg_set_tuning(37, 1)
    assert grads[0][0] == grads[1][0]
    # the whole gradient vector, and every tensor that is not itself rounding noise (a 16-bit tape leaves ~1e-2 of a layer's
    # typical gradient as noise on the tensors whose true gradient nearly cancels)
    keys = list(grads[0][1])
    flat = {on: torch.cat([grads[on][1][k].reshape(-1).double() for k in keys]) for on in (0, 1)}
    whole = float((flat[1] - flat[0]).norm() / flat[0].norm())
    norms = torch.tensor([float(grads[0][1][k].norm()) / max(1, grads[0][1][k].numel()) ** 0.5 for k in keys])
    floor = float(norms.median()) * 0.1
    worst, worst_key, differ = 0.0, None, 0
    for k, rms in zip(keys, norms.tolist()):
        differ += int(not torch.equal(grads[1][1][k], grads[0][1][k]))
        if rms >= floor:
            r = _rel(grads[1][1][k], grads[0][1][k])
            if r > worst:
                worst, worst_key = r, k
    assert differ > 0, "key 37 changed nothing: no layer of this net took the GNB kernel at this batch"
    assert whole <= (1e-4 if dtn == "fp32" else 5e-3), whole
    assert worst <= (2e-4 if dtn == "fp32" else 2e-2), (worst, worst_key)


def test_query_and_dispatch_agree_over_a_shape_sweep(gnb_workgroups):
    """`dsg_conv2d_gnb_supported` mirrors the launcher's kernel selection (workgroup shape by grid size, 32-cout workgroups for
    small grids, the 128-cout tiles' straddle rule): over a sweep of batch / channel / map sizes every call the query accepts must
    run (the launcher refuses a gnb call that did not end in a GNB kernel) and return the sums of what it wrote -- checked through
    sum du per (n, c) against torch on the stored dA -- and every call it rejects must raise."""
    g = lambda t: t.to(DEV)
    taken = refused = 0
    for dtn in ("bf16", "fp32"):
        dt = ops.dtype_code(dtn)
        blocked = dtn != "fp32"
        for (n, cdy, c0, c1, h, w) in [(1, 64, 64, 0, 32, 32), (2, 64, 64, 0, 64, 64), (4, 64, 64, 0, 64, 64), (16, 64, 64, 0, 64, 64),
                                       (2, 128, 128, 0, 32, 64), (8, 128, 128, 0, 32, 32), (32, 128, 128, 0, 32, 32), (3, 64, 192, 0, 64, 32),
                                       (8, 64, 64, 64, 64, 64), (8, 64, 128, 128, 32, 32), (2, 64, 256, 128, 32, 32), (8, 64, 320, 0, 32, 32),
                                       (5, 32, 96, 0, 64, 64), (1, 256, 512, 0, 32, 32), (9, 64, 64, 0, 8, 32),
                                       (8, 64, 192, 64, 64, 64), (16, 32, 64, 32, 64, 64), (8, 64, 96, 32, 64, 64),
                                       (32, 64, 64, 64, 64, 64), (8, 64, 112, 16, 64, 64), (4, 64, 32, 96, 64, 64)]:
            x, dy, wt, gamma, beta = _case(n, cdy, c0, c1, h, w, True, 11 * n + c0 + h)
            c = c0 + c1
            if blocked:
                xb = ops.to_blocked(g(x), dt)
                x0b, x1b = xb[:, :c0 // 8].contiguous(), (xb[:, c0 // 8:].contiguous() if c1 else None)
                dyb, xr = ops.to_blocked(g(dy), dt), ops.from_blocked(xb).cpu()
                whd = ops.pack_conv_weight(g(wt), ops.PACK_DGRAD, dt)
            else:
                x0b, x1b, dyb, xr = g(x[:, :c0]).contiguous(), (g(x[:, c0:]).contiguous() if c1 else None), g(dy), x
                whd = ops.relayout_conv_weight_h2_dgrad(g(wt))
            ss, _ = ops.gn_scale_shift_train(g(xr[:, :c0]).contiguous(), g(gamma), g(beta), GROUPS, 1e-5,
                                             src1=g(xr[:, c0:]).contiguous() if c1 else None)
            kw = dict(ksize=3, cout=c, src_blocked=blocked, dst_blocked=blocked, compute_dtype=dt, weight_h2=whd)
            if blocked:
                kw["weight_h2_stride"] = (c + 63) // 64 * 64
            wd = ops.relayout_conv_weight_dgrad(g(wt))
            gnb = dict(x0=x0b, x1=x1b, ss=ss, silu=True)
            if ops.conv2d_fused(dyb, wd, gnb=dict(gnb, query_only=True), **kw):
                da, parts = ops.conv2d_fused(dyb, wd, gnb=gnb, want_stats=True, **kw)
                daf = (ops.from_blocked(da) if blocked else da).cpu().double()
                u = xr.double() * ss[:, :, 0].cpu().double()[:, :, None, None] + ss[:, :, 1].cpu().double()[:, :, None, None]
                s = torch.sigmoid(u)
                want = (daf * (s * (1 + u * (1 - s)))).sum((2, 3))
                got = parts.sum(2)[..., 0].cpu()
                assert _rel(got, want) <= 2e-5, (dtn, n, c0, c1, h, w, _rel(got, want))
                taken += 1
            else:
                with pytest.raises(RuntimeError):
                    ops.conv2d_fused(dyb, wd, gnb=gnb, want_stats=True, **kw)
                refused += 1
    assert taken >= 10 and refused >= 6, (taken, refused)
