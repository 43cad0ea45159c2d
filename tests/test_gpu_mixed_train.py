"""Mixed-precision training tape (bf16 / fp16): kernels of the backward walk on 16-bit channel-blocked tensors, one
whole training step against torch-CPU autograd of the fp32 oracle, GradScaler semantics (a non-finite step is skipped),
and the fp32 tape's internal loss scale (tiny dy through the fp16x2 split).

Reference: DriveSceneGen/scripts/train.py:24 (mixed_precision='fp16'), pipeline/training_pipeline.py:48-49,84-91;
BASELINE.json configs[4] (bf16)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import ops, synth  # noqa: E402
from oracle.scheduler_oracle import OracleDDPMScheduler  # noqa: E402
from oracle.unet_oracle import OracleUNet2DModel  # noqa: E402
from tests.common import CFG1, CFG4_SMALL, rel_l2, synth_weights  # noqa: E402

DEV = "cuda"
TDT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


def _rnd(x, mode):
    return x.to(TDT[mode]).float()


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [("gn_silu", 64, 0, 64, 32, 64, 2, True, 3), ("concat", 128, 64, 128, 16, 32, 3, True, 3),
                                  ("plain_wide", 64, 0, 192, 8, 96, 2, False, 3),
                                  ("two_co_blocks_of_128", 64, 0, 256, 8, 64, 5, True, 3),
                                  ("plain_128_couts", 192, 0, 128, 4, 32, 2, False, 3),
                                  ("pointwise_shortcut", 128, 64, 64, 16, 32, 3, False, 1),
                                  ("pointwise_odd_rows", 64, 0, 128, 6, 32, 2, True, 1),
                                  ("pointwise_128x128_tiles", 128, 128, 256, 8, 32, 3, True, 1),
                                  ("pointwise_128x64_tiles", 256, 0, 64, 16, 32, 2, False, 1)], ids=lambda c: c[0])
def test_wgrad16(case, mode):
    """dsg_conv2d_wgrad on blocked 16-bit x / dY (transposing LDS reads) vs an fp64 evaluation of the same rounded
    operands; 3x3 and pointwise."""
    _, c0, c1, cout, h, w, n, gn, k = case
    cin = c0 + c1
    x0, x1 = _rnd(_t(1, (n, c0, h, w)), mode), (_rnd(_t(2, (n, c1, h, w)), mode) if c1 else None)
    dy = _rnd(_t(3, (n, cout, h, w), 0.3), mode)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    b0, b1 = ops.to_blocked(x0.to(DEV), mode), (ops.to_blocked(x1.to(DEV), mode) if c1 else None)
    act, ss = xin.double(), None
    if gn:
        gamma, beta = 1 + _t(5, (cin,), 0.1), _t(6, (cin,), 0.1)
        ss = ops.gn_scale_shift_from_parts(ops.gn_channel_stats_blocked(b0), gamma.to(DEV), beta.to(DEV), 32, 1e-5, h * w,
                                           stats1=ops.gn_channel_stats_blocked(b1) if c1 else None)
        act = F.silu(xin.double() * ss.cpu()[:, :, 0, None, None].double() + ss.cpu()[:, :, 1, None, None].double())
    act = _rnd(act.float(), mode).double()
    ref = torch.nn.grad.conv2d_weight(act, (cout, cin, k, k), dy.double(), padding=k // 2)
    dw = torch.full((cout, cin, k, k), 0.25, dtype=torch.float32, device=DEV)   # accumulated into
    sums = torch.full((n, cout + 3), -7.0, dtype=torch.float32, device=DEV)   # by-product: per-(n, cout) sums of dY
    ops.conv_wgrad(b0, ops.to_blocked(dy.to(DEV), mode), dw, src1=b1, ksize=k, gn_scale_shift=ss, silu=gn,
                   dy_sums=sums[:, 1:], dy_sums_stride=sums.stride(0))
    assert torch.allclose(sums[:, 1:cout + 1].cpu().double(), dy.double().sum((2, 3)), rtol=1e-5, atol=1e-4)
    assert float(sums[:, 0].min()) == -7.0 and float(sums[:, cout + 1:].max()) == -7.0   # nothing outside the window
    got = dw.cpu().double() - 0.25
    assert rel_l2(got, ref) <= (2e-4 if gn else 2e-5), rel_l2(got, ref)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [("upsample_wide", "ups", 128, 128, 16, 32, 3), ("upsample_64_couts", "ups", 64, 64, 8, 64, 2),
                                  ("upsample_two_strips", "ups", 64, 256, 6, 64, 2), ("upsample_batch_24", "ups", 64, 128, 8, 32, 24),
                                  ("upsample_512_wide", "ups", 256, 512, 4, 32, 1), ("upsample_192_couts", "ups", 128, 192, 16, 32, 2),
                                  ("stride2_wide", "s2", 128, 128, 32, 64, 3), ("stride2_64_couts", "s2", 64, 64, 16, 128, 2)],
                         ids=lambda c: c[0])
def test_wgrad16_sampler_convs_read_the_half_resolution_operand_in_place(case, mode):
    """The two sampler convs of the U-Net on the 16-bit weight-gradient kernel without a materialised copy (round 6):
    Upsample2D's conv (train.py:39-57's up blocks: nearest x2 then 3x3) reads the LOW-resolution x -- the tape used to write a
    4x larger upsampled tensor first; Downsample2D's stride-2 conv takes dY as zero between its pixels -- the tape used to copy
    dY into a zeroed full-resolution buffer with a strided torch copy.  Against fp64 on the same rounded operands; the nine-tap
    forms (x at (y >> 1, x >> 1): dsg_set_tuning key 39 = 0 for the up-sampler) bitwise against the old route (the same kernel
    on the materialised tensors: the products and their order are the same); the up-sampler's folded form (key 39 = 1, the
    default: x's own map as the K grid, dY as its space-to-depth image, 4 taps per pixel parity) against fp64 and the nine-tap
    form."""
    from drivescenegen_amd import _lib
    _, kind, cin, cout, h, w, n = case          # (h, w): x's map
    x = _rnd(_t(11, (n, cin, h, w)), mode)
    xb = ops.to_blocked(x.to(DEV), mode)
    dw = torch.full((cout, cin, 3, 3), 0.5, dtype=torch.float32, device=DEV)
    old = torch.full((cout, cin, 3, 3), 0.5, dtype=torch.float32, device=DEV)
    sums, sums_old = (torch.zeros((n, cout), dtype=torch.float32, device=DEV) for _ in range(2))
    if kind == "ups":
        dy = _rnd(_t(12, (n, cout, 2 * h, 2 * w), 0.3), mode)
        dyb = ops.to_blocked(dy.to(DEV), mode)
        ref = torch.nn.grad.conv2d_weight(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), (cout, cin, 3, 3),
                                          dy.double(), padding=1)
        assert ops.wgrad16_supported(cin, 0, cout, h, w, 3, 1, True)
        try:
            _lib.check(_lib.load().dsg_set_tuning(39, 0))
            ops.conv_wgrad(xb, dyb, dw, ksize=3, upsample=True, dy_sums=sums)
        finally:
            _lib.load().dsg_set_tuning(39, 1)
        ops.conv_wgrad(ops.upsample_nearest2x(xb), dyb, old, ksize=3, dy_sums=sums_old)
        # the folded form: the same sums in another order
        fdw = torch.full((cout, cin, 3, 3), 0.5, dtype=torch.float32, device=DEV)
        fsums = torch.full((n, cout + 2), -3.0, dtype=torch.float32, device=DEV)
        fbias = torch.full((cout,), 0.25, dtype=torch.float32, device=DEV)
        ops.conv_wgrad(xb, dyb, fdw, ksize=3, upsample=True, dy_sums=fsums[:, 1:], dy_sums_stride=fsums.stride(0), bias_grad=fbias)
        assert rel_l2(fdw.cpu().double() - 0.5, ref) <= 2e-5, rel_l2(fdw.cpu().double() - 0.5, ref)
        assert float((fdw - dw).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6 * float((dw - 0.5).abs().max())
        assert torch.allclose(fsums[:, 1:cout + 1].cpu().double(), dy.double().sum((2, 3)), rtol=1e-5, atol=1e-4)
        assert float(fsums[:, 0].min()) == -3.0 and float(fsums[:, cout + 1:].max()) == -3.0
        assert torch.allclose(fbias.cpu().double() - 0.25, dy.double().sum((0, 2, 3)), rtol=1e-5, atol=2e-4)
    else:
        dy = _rnd(_t(12, (n, cout, h // 2, w // 2), 0.3), mode)
        dyb = ops.to_blocked(dy.to(DEV), mode)
        ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), dy.double(), stride=2, padding=1)
        assert ops.wgrad16_supported(cin, 0, cout, h, w, 3, 2, False)
        ops.conv_wgrad(xb, dyb, dw, ksize=3, stride=2, dy_sums=sums)
        up = torch.zeros((n, cout // 8, h, w, 8), dtype=dyb.dtype, device=DEV)
        up[:, :, ::2, ::2].copy_(dyb)
        ops.conv_wgrad(xb, up, old, ksize=3, dy_sums=sums_old)
    assert torch.equal(dw, old) and torch.equal(sums, sums_old)
    assert torch.allclose(sums.cpu().double(), dy.double().sum((2, 3)), rtol=1e-5, atol=1e-4)
    assert rel_l2(dw.cpu().double() - 0.5, ref) <= 2e-5, rel_l2(dw.cpu().double() - 0.5, ref)
    # a shape outside the sampler form says so
    assert not ops.wgrad16_supported(cin, 64, cout, h, w, 3, 1, True) and not ops.wgrad16_supported(cin, 0, cout, 8, 16, 3, 2, False)


@pytest.mark.parametrize("act", ["plain", "gn_silu", "gn_affine"])
@pytest.mark.parametrize("shape", [(128, 64, 128, 16, 32, 3), (64, 0, 256, 12, 64, 5), (384, 0, 128, 32, 32, 4)],
                         ids=["concat_192_128", "64_256_two_strips", "384_128"])
def test_wgrad16_wide_workgroups_match_the_64x64_ones(shape, act):
    """cout % 128 == 0 selects 64 ci x 128 co workgroups (a wave keeps two co tiles, one workgroup per CU; dsg_set_tuning
    key 29): the same products in the same order per run, other runs of pixels per partial slab -- equal to fp32 round-off
    of the slab sum, and the dY sums likewise."""
    from drivescenegen_amd import _lib
    c0, c1, cout, h, w, n = shape
    cin = c0 + c1
    b0 = ops.to_blocked(_t(21, (n, c0, h, w)).to(DEV), "bf16")
    b1 = ops.to_blocked(_t(22, (n, c1, h, w)).to(DEV), "bf16") if c1 else None
    dy = ops.to_blocked(_t(23, (n, cout, h, w), 0.3).to(DEV), "bf16")
    ss = None
    if act != "plain":
        ss = torch.stack([1 + _t(24, (n, cin), 0.1), _t(25, (n, cin), 0.1)], -1).contiguous().to(DEV)
    res = []
    try:
        for wide in (0, 1):
            _lib.check(_lib.load().dsg_set_tuning(29, wide))
            dw = torch.zeros((cout, cin, 3, 3), dtype=torch.float32, device=DEV)
            sums = torch.zeros((n, cout), dtype=torch.float32, device=DEV)
            bg = torch.zeros(cout, dtype=torch.float32, device=DEV)
            ops.conv_wgrad(b0, dy, dw, src1=b1, ksize=3, gn_scale_shift=ss, silu=act == "gn_silu", dy_sums=sums, bias_grad=bg)
            res.append((dw, sums, bg))
    finally:
        _lib.load().dsg_set_tuning(29, 1)
    for a, b in zip(*res):
        assert torch.isfinite(b).all()
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()), float((a - b).abs().max())


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_wgrad_finishes_the_bias_gradient_with_the_dy_sums(mode):
    """dsg_conv_wgrad_args.dy_bias_grad: the pass that turns the per-run sums of dY into per-(n, cout) sums also adds their
    sum over the batch to the bias gradient (one dsg_reduce_rows_add launch per conv less)."""
    n, c, cout, h, w = 5, 64, 128, 16, 32
    dy = _t(3, (n, cout, h, w), 0.3)
    x = _t(1, (n, c, h, w))
    if mode == "fp32":
        xs, dys = x.to(DEV), dy.to(DEV)
    else:
        dy, x = _rnd(dy, mode), _rnd(x, mode)
        xs, dys = ops.to_blocked(x.to(DEV), mode), ops.to_blocked(dy.to(DEV), mode)
    dw = torch.zeros((cout, c, 3, 3), dtype=torch.float32, device=DEV)
    sums = torch.zeros((n, cout), dtype=torch.float32, device=DEV)
    bg = torch.full((cout,), 0.5, dtype=torch.float32, device=DEV)
    ops.conv_wgrad(xs, dys, dw, ksize=3, dy_sums=sums, bias_grad=bg)
    ref = dy.double().sum((2, 3))
    assert torch.allclose(sums.cpu().double(), ref, rtol=1e-5, atol=1e-4)
    assert torch.allclose(bg.cpu().double(), 0.5 + ref.sum(0), rtol=1e-5, atol=1e-4)
    # and the weight gradient is what it is without the by-products
    dw2 = torch.zeros_like(dw)
    ops.conv_wgrad(xs, dys, dw2, ksize=3)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_streaming_backward_ops_blocked16(mode):
    n, c0, c1, h, w = 2, 32, 16, 16, 32
    x0, x1 = _rnd(_t(11, (n, c0, h, w)) + 0.3, mode), _rnd(_t(12, (n, c1, h, w)) * 2, mode)
    dy = _rnd(_t(13, (n, c0 + c1, h, w), 0.2), mode)
    add0, add1 = _rnd(_t(14, (n, c0, h, w), 0.1), mode), _rnd(_t(15, (n, c1, h, w), 0.1), mode)
    gamma, beta = (1 + _t(16, (c0 + c1,), 0.1)).to(DEV), _t(17, (c0 + c1,), 0.1).to(DEV)
    dev = lambda t: t.to(DEV)
    blk = lambda t: ops.to_blocked(t.to(DEV), mode)
    ss, mr = ops.gn_scale_shift_train(dev(x0), gamma, beta, 8, 1e-5, src1=dev(x1))
    for silu in (False, True):
        dg_ref, db_ref = torch.zeros(c0 + c1, device=DEV), torch.zeros(c0 + c1, device=DEV)
        r0, r1 = ops.gn_bwd(dev(x0), dev(dy), ss, mr, gamma, 8, silu, dg_ref, db_ref, src1=dev(x1), add0=dev(add0), add1=dev(add1))
        dg, db = torch.zeros(c0 + c1, device=DEV), torch.zeros(c0 + c1, device=DEV)
        g0, g1 = ops.gn_bwd_blocked(blk(x0), blk(dy), ss, mr, gamma, 8, silu, dg, db, src1=blk(x1), add0=blk(add0), add1=blk(add1))
        tol = 4e-3 if mode == "bf16" else 6e-4   # one rounding of the 16-bit result
        assert rel_l2(ops.from_blocked(g0).cpu(), r0.cpu()) <= tol and rel_l2(ops.from_blocked(g1).cpu(), r1.cpu()) <= tol
        assert rel_l2(dg.cpu(), dg_ref.cpu()) <= 1e-5 and rel_l2(db.cpu(), db_ref.cpu()) <= 1e-5
    # per-channel sums (bias gradients), fan-in add, nearest x2 and its adjoint
    assert torch.allclose(ops.channel_sums(blk(dy)).cpu(), dy.sum((2, 3)), rtol=1e-5, atol=1e-4)
    assert torch.equal(ops.from_blocked(ops.add(blk(x0), blk(add0))).cpu(), _rnd(x0 + add0, mode))
    up = ops.upsample_nearest2x(blk(x0))
    assert torch.equal(ops.from_blocked(up).cpu(), F.interpolate(x0, scale_factor=2.0, mode="nearest"))
    big = _rnd(_t(18, (n, c0, 2 * h, 2 * w)), mode)
    pooled = ops.from_blocked(ops.sumpool2x2(blk(big), add=blk(add0))).cpu()
    assert torch.equal(pooled, _rnd(F.avg_pool2d(big, 2) * 4 + add0, mode))


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_dgrad_of_stride2_conv_as_phase_convs(mode):
    """The data gradient of Downsample2D's conv (3x3, stride 2) through the folded up-sampler kernel with the
    PACK_DGRAD_S2 operand image, against conv_transpose2d on the rounded operands."""
    n, cin, cout, h, w = 2, 64, 64, 32, 64            # dY is [n, cout, 16, 32]
    wt = _t(21, (cout, cin, 3, 3), 1.0 / 24)
    dy = _rnd(_t(22, (n, cout, h // 2, w // 2)), mode)
    prev = _rnd(_t(23, (n, cin, h, w), 0.1), mode)
    ref = F.conv_transpose2d(dy.double(), _rnd(wt, mode).double(), stride=2, padding=1, output_padding=1) + prev.double()
    wd = ops.relayout_conv_weight_dgrad(wt.to(DEV))
    got = ops.conv2d_fused(ops.to_blocked(dy.to(DEV), mode), wd, ksize=3, upsample=True, cout=cin,
                           residual=ops.to_blocked(prev.to(DEV), mode), src_blocked=True, dst_blocked=True,
                           compute_dtype=mode, weight_h2_fold=ops.pack_conv_weight(wt.to(DEV), ops.PACK_DGRAD_S2, mode))
    assert rel_l2(ops.from_blocked(got).cpu(), ref) <= (3e-3 if mode == "bf16" else 8e-4)


def _oracle_step(cfg, x0, noise, t, loss_mult=1.0):
    ora = synth_weights(OracleUNet2DModel(**cfg)).train()
    noisy = OracleDDPMScheduler().add_noise(x0, noise, t)
    loss = F.mse_loss(ora(noisy, t, return_dict=False)[0], noise) * loss_mult
    loss.backward()
    return ora, noisy, float(loss.detach())


def _grad_report(net, ora):
    og = dict(ora.named_parameters())
    num = den = 0.0
    worst = ("", 0.0)
    for name, p in net.named_parameters():
        g, w = p.grad.detach().cpu().double(), og[name].grad.double()
        num += float((g - w).pow(2).sum())
        den += float(w.pow(2).sum())
        if float(w.norm()) > 1e-3 * (den ** 0.5 + 1e-30):   # tensors that matter for the update direction
            e = rel_l2(g, w)
            if e > worst[1]:
                worst = (name, e)
    return (num / den) ** 0.5, worst


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("cfg", [CFG1, CFG4_SMALL], ids=["cfg1_tiny", "attn_blocks"])
def test_training_step_mixed_vs_fp32_oracle(cfg, mode):
    """One DDPM training step in mixed precision vs torch-CPU autograd of the fp32 oracle: loss within 1e-2 relative,
    all 16-bit-class gradients: global rel-L2 over the whole gradient vector <= 3e-2 (measured 5e-3), worst significant tensor <= 1.2e-1."""
    b, ss = 3, cfg["sample_size"]
    x0 = torch.from_numpy(synth.synth_scene_rasters(b, cfg["in_channels"], ss, ss, 5))
    noise = torch.from_numpy(synth.normal(6, tuple(x0.shape)))
    t = torch.tensor([12, 500, 987])
    ora, noisy, loss_o = _oracle_step(cfg, x0, noise, t)
    net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).train().set_compute_dtype(mode)
    pred = net(noisy.to(DEV), t.to(DEV), return_dict=False)[0]
    assert pred.dtype == torch.float32 and pred.requires_grad
    loss = d.mse_loss(pred, noise.to(DEV))
    loss.backward()
    assert abs(float(loss.detach().cpu()) - loss_o) <= 1e-2 * loss_o
    total, worst = _grad_report(net, ora)
    assert total <= 3e-2 and worst[1] <= 1.2e-1, (total, worst)


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
@pytest.mark.parametrize("cfg", [CFG1, CFG4_SMALL], ids=["cfg1_tiny", "attn_blocks"])
def test_upsampler_data_gradient_routes_agree(cfg, mode, monkeypatch):
    """The 16-bit tape's up-sampler data gradient: one 4x4 stride-2 window over dY (dsg_conv_args.s2_window4, the route the
    tape takes) against the 3x3 data gradient at full resolution + 2x2 sums it replaces (DSG_UPS_DGRAD_FULLRES=1): the same
    sums with two roundings less -- whole gradient vector within 1e-2 of each other (fp32 tape, [N,C,H,W] tensors: 2e-6), loss
    bit-identical (same forward)."""
    from drivescenegen_amd import autograd as ag
    b, ss = 2, cfg["sample_size"]
    x0 = torch.from_numpy(synth.synth_scene_rasters(b, cfg["in_channels"], ss, ss, 5))
    noise = torch.from_numpy(synth.normal(6, tuple(x0.shape)))
    t = torch.tensor([12, 700])
    calls = []
    real = ops.conv2d_fused

    def spy(*a, **kw):
        calls.append(bool(kw.get("s2_window4")))
        return real(*a, **kw)

    monkeypatch.setattr(ops, "conv2d_fused", spy)
    res = []
    for fullres in (False, True):
        monkeypatch.setattr(ag, "_UPS_DGRAD_FULLRES", fullres)
        calls.clear()
        net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).train().set_compute_dtype(mode)
        sch = d.DDPMScheduler()
        noisy = sch.add_noise(x0.to(DEV), noise.to(DEV), t.to(DEV))
        loss = d.mse_loss(net(noisy, t.to(DEV), return_dict=False)[0], noise.to(DEV))
        loss.backward()
        n_ups = sum(1 for k in net.state_dict() if k.endswith("upsamplers.0.conv.weight"))
        assert sum(calls) == (0 if fullres else n_ups), (sum(calls), n_ups)
        res.append((float(loss.detach().cpu()), {k: p.grad.detach().cpu().double() for k, p in net.named_parameters()}))
    assert res[0][0] == res[1][0]
    num = sum(float((res[0][1][k] - res[1][1][k]).pow(2).sum()) for k in res[0][1])
    den = sum(float(res[1][1][k].pow(2).sum()) for k in res[0][1])
    assert (num / den) ** 0.5 <= (1e-2 if mode == "bf16" else 2e-6), (num / den) ** 0.5


def test_folded_upsampler_weight_gradient_in_a_whole_step():
    """dsg_set_tuning key 39 on / off on a whole bf16 training step of the tiny network (its up-sampler conv -- 64 -> 64 channels
    on a 32 x 32 map -- takes the folded kernel): same forward, every gradient within fp32 round-off of the other form (the
    two forms add the same bf16 products in another order)."""
    from drivescenegen_amd import _lib
    cfg = CFG1
    b, ss = 2, cfg["sample_size"]
    x0 = torch.from_numpy(synth.synth_scene_rasters(b, cfg["in_channels"], ss, ss, 5))
    noise = torch.from_numpy(synth.normal(6, tuple(x0.shape)))
    t = torch.tensor([12, 700])
    res = []
    try:
        for fold in (1, 0):
            _lib.check(_lib.load().dsg_set_tuning(39, fold))
            net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).train().set_compute_dtype("bf16")
            noisy = d.DDPMScheduler().add_noise(x0.to(DEV), noise.to(DEV), t.to(DEV))
            loss = d.mse_loss(net(noisy, t.to(DEV), return_dict=False)[0], noise.to(DEV))
            loss.backward()
            res.append((float(loss.detach().cpu()), {k: p.grad.detach().cpu().double() for k, p in net.named_parameters()}))
    finally:
        _lib.load().dsg_set_tuning(39, 1)
    assert res[0][0] == res[1][0]
    ups = [k for k in res[0][1] if k.endswith("upsamplers.0.conv.weight")]
    assert ups
    for k in res[0][1]:
        a, b2 = res[0][1][k], res[1][1][k]
        if k in ups or k.endswith("upsamplers.0.conv.bias"):
            assert float((a - b2).abs().max()) <= 2e-5 * float(b2.abs().max()) + 1e-12, k
        else:
            assert torch.equal(a, b2), k


def test_fp16_grad_scaler_skips_a_non_finite_step_and_recovers():
    """accelerate's fp16 path (training_pipeline.py:86-91 under train.py:24): scaled backward, unscale + clip, a step
    whose gradients are not finite is skipped -- parameters, AdamW moments and the LR schedule stay put, the scale
    halves -- and the next clean step goes through."""
    acc = d.Accelerator(mixed_precision="fp16")
    net = synth_weights(d.UNet2DModel(**CFG1))
    opt = d.AdamW(net.parameters(), lr=1e-3)
    lrs = d.get_cosine_schedule_with_warmup(optimizer=opt, num_warmup_steps=0, num_training_steps=100)
    net, popt, plrs = acc.prepare(net, opt, lrs)
    assert net.compute_dtype == "fp16" and acc.scaler.get_scale() == 65536.0
    x = torch.from_numpy(synth.normal(3, (2, 3, 64, 64))).to(DEV)
    t = torch.tensor([10, 700], device=DEV)

    def one(inp):
        with acc.accumulate(net):
            loss = d.mse_loss(net(inp, t, return_dict=False)[0], torch.zeros_like(inp))
            acc.backward(loss)
            acc.clip_grad_norm_(net.parameters(), 1.0)
            popt.step()
            plrs.step()
            popt.zero_grad()
        return loss
    one(x)                                   # clean step: creates the flat slabs, moves the weights
    assert not popt.step_was_skipped and acc.scaler.get_scale() == 65536.0
    before = [p.detach().clone() for p in net.parameters()]
    lr_before, step_before = plrs.get_last_lr()[0], list(opt._flat.values())[0]["step"]
    bad = x.clone()
    bad[0, 0, 0, 0] = float("inf")
    one(bad)
    assert popt.step_was_skipped and acc.scaler.get_scale() == 32768.0
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, net.parameters()))
    assert plrs.get_last_lr()[0] == lr_before and list(opt._flat.values())[0]["step"] == step_before
    assert all(float(p.grad.abs().max()) == 0.0 for p in net.parameters())   # zero_grad still ran
    one(x)
    assert not popt.step_was_skipped and acc.scaler.get_scale() == 32768.0
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, net.parameters()))
    assert all(torch.isfinite(p).all() for p in net.parameters())


def test_bf16_accelerator_needs_no_scaler_and_accumulates():
    """bf16: no GradScaler; with gradient_accumulation_steps = 2 the optimizer moves on every second micro-batch only
    and the accumulated gradient equals that of the two micro-batches (prepared optimizer)."""
    acc = d.Accelerator(mixed_precision="bf16", gradient_accumulation_steps=2)
    net = synth_weights(d.UNet2DModel(**CFG1))
    opt = d.AdamW(net.parameters(), lr=1e-3)
    net, popt = acc.prepare(net, opt)
    assert acc.scaler is None and net.compute_dtype == "bf16"
    xs = [torch.from_numpy(synth.normal(40 + i, (2, 3, 64, 64))).to(DEV) for i in range(2)]
    t = torch.tensor([10, 700], device=DEV)
    w0 = net.conv_in.weight.detach().clone()
    grads = []
    for i, x in enumerate(xs):
        with acc.accumulate(net):
            acc.backward(d.mse_loss(net(x, t, return_dict=False)[0], torch.zeros_like(x)))
            grads.append(net.conv_out.weight.grad.detach().clone())
            popt.step()
            popt.zero_grad()
        assert torch.equal(w0, net.conv_in.weight.detach()) == (i == 0)
    # micro-batch losses are divided by the accumulation count; the slab held g0/2 after the first, (g0 + g1)/2 after the second
    solo = synth_weights(d.UNet2DModel(**CFG1)).to(DEV).train().set_compute_dtype("bf16")
    d.mse_loss(solo(xs[1], t, return_dict=False)[0], torch.zeros_like(xs[1])).backward()
    want = grads[0] + 0.5 * solo.conv_out.weight.grad
    assert rel_l2(grads[1].cpu(), want.cpu()) <= 1e-5


def test_fp32_tape_keeps_precision_when_dy_is_tiny():
    """at production size (256^2, B = 16) the mean-reduced loss gives |dy| ~ 1e-7, below the range where the
    fp16 pairs of the split path keep fp32 precision.  The tape scales its walk by a power of two and takes it back out
    of the slab: gradients of a loss shrunk to that regime still match fp32 autograd at the usual tolerance."""
    cfg, b = CFG1, 3
    x0 = torch.from_numpy(synth.synth_scene_rasters(b, 3, 64, 64, 5))
    noise = torch.from_numpy(synth.normal(6, tuple(x0.shape)))
    t = torch.tensor([12, 500, 987])
    shrink = float(x0.numel()) / (16 * 4 * 256 * 256)          # dy as small as at 256^2 x 4 channels x B = 16
    ora, noisy, _ = _oracle_step(cfg, x0, noise, t, loss_mult=shrink)
    net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).train()
    loss = d.mse_loss(net(noisy.to(DEV), t.to(DEV), return_dict=False)[0], noise.to(DEV))
    (loss * shrink).backward()
    og = dict(ora.named_parameters())
    top = max(float(w.grad.abs().max()) for w in og.values())
    bad = []
    for name, p in net.named_parameters():
        g, w = p.grad.detach().cpu(), og[name].grad
        # (gradients that are zero in exact arithmetic -- a bias in front of a one-channel-per-group norm -- are
        #  round-off on both sides: they are held to an absolute bound relative to the largest gradient)
        if float((g - w).abs().max()) > 2e-6 * top and rel_l2(g, w) > 2e-4:
            bad.append((name, rel_l2(g, w), float((g - w).abs().max()) / top))
    assert not bad, bad[:6]
