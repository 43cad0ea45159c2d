"""Mixed-precision (bf16 / fp16) kernels through the C ABI -- BASELINE.json configs[4] and the reference's own
``mixed_precision='fp16'`` (train.py:24, training_pipeline.py:48-49).

Per-op references are torch-CPU fp64 evaluations of the SAME rounded operands the kernel multiplies (16-bit rounded
sources and weights, fp32 GroupNorm affine + SiLU rounded once more, exact products, wide accumulation), so what is
left is the accumulation order and the final rounding of a 16-bit result: 2^-9 relative for bf16, 2^-11 for fp16 per
element (rel-L2 bounds 3e-3 / 8e-4); fp32 results within 2e-5.  Whole-network bound: rel-L2 <= 2e-2 against the
fp32 oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import ops, synth  # noqa: E402
from tests.common import CFG1, CFG4_SMALL, CFG5, noisy_inputs, rel_l2, synth_weights  # noqa: E402

DEV = "cuda"
TDT = {"bf16": torch.bfloat16, "fp16": torch.float16}
OUT_TOL = {"bf16": 3e-3, "fp16": 8e-4}


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


def _rnd(x, mode):
    return x.to(TDT[mode]).float()


def _blk(x, mode):
    """fp32 [N,C,H,W] CPU -> 16-bit channel-blocked device tensor"""
    return ops.to_blocked(x.to(DEV), mode)


MIX_CASES = [
    # name, c0, c1, cout, h, w, k, stride, ups, gn, temb, res, batch
    ("res3x3_gn_temb", 64, 0, 64, 32, 64, 3, 1, False, True, True, False, 3),
    ("res3x3_resid_16rows", 64, 0, 128, 32, 32, 3, 1, False, True, False, True, 64),     # 256 16-row workgroups: NT = 4
    ("concat_straddle", 128, 64, 128, 16, 32, 3, 1, False, True, True, False, 2),
    ("plain3x3_noact", 32, 0, 64, 16, 32, 3, 1, False, False, False, True, 2),
    ("narrow_16x16", 64, 0, 64, 16, 16, 3, 1, False, True, True, True, 2),
    ("upsample_fold", 64, 0, 64, 16, 32, 3, 1, True, False, False, False, 2),
    ("upsample_fold_16rows", 128, 0, 128, 32, 32, 3, 1, True, False, False, False, 16),
    ("stride2", 64, 0, 64, 32, 64, 3, 2, False, False, False, False, 2),
    ("stride2_16rows", 64, 0, 128, 64, 64, 3, 2, False, False, False, False, 64),
    ("shortcut_1x1", 128, 64, 64, 16, 32, 1, 1, False, False, False, False, 2),
    ("small_grid_bm32", 256, 0, 256, 8, 32, 3, 1, False, True, True, True, 1),
    ("cout128_tiles_16rows", 128, 0, 128, 32, 32, 3, 1, False, True, True, True, 128),   # 128-cout workgroups, NT = 4
    ("cout128_tiles_8rows", 64, 0, 256, 8, 64, 3, 1, False, True, False, True, 64),      # ... NT = 2
]


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("case", MIX_CASES, ids=[c[0] for c in MIX_CASES])
def test_conv_blocked16(case, mode):
    name, c0, c1, cout, h, w, k, stride, ups, gn, temb, res, batch = case
    cin = c0 + c1
    x0, x1 = _rnd(_t(1, (batch, c0, h, w)), mode), (_rnd(_t(2, (batch, c1, h, w)), mode) if c1 else None)
    wt = _t(3, (cout, cin, k, k), 1.0 / np.sqrt(cin * k * k))
    bias = _t(4, (cout,), 0.1)
    gamma, beta = 1 + _t(5, (cin,), 0.1), _t(6, (cin,), 0.1)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    dv = lambda t: None if t is None else t.to(DEV)
    b0, b1 = _blk(x0, mode), (_blk(x1, mode) if c1 else None)
    ss = None
    act = xin.double()
    if gn:
        groups = 32 if cin % 32 == 0 else 8
        st0 = ops.gn_channel_stats_blocked(b0, splits=2)
        st1 = ops.gn_channel_stats_blocked(b1, splits=1) if c1 else None
        ss = ops.gn_scale_shift_from_parts(st0, dv(gamma), dv(beta), groups, 1e-5, h * w, stats1=st1)
        ref_gn = F.group_norm(xin.double(), groups, gamma.double(), beta.double(), 1e-5)
        got_gn = xin.double() * ss.cpu()[:, :, 0, None, None].double() + ss.cpu()[:, :, 1, None, None].double()
        assert rel_l2(got_gn, ref_gn) <= 1e-5   # statistics of the 16-bit tensor, fp64 partial sums
        act = F.silu(got_gn)
    act = _rnd(act.float(), mode).double()       # the operand is rounded once, after the fp32 affine + SiLU
    if ups:
        act = F.interpolate(act, scale_factor=2.0, mode="nearest")
    if ups:   # the folded form sums taps in fp32 BEFORE rounding the weight: reference = the four phase convs
        wq = None
    else:
        wq = _rnd(wt, mode).double()
    if wq is not None:
        ref = F.conv2d(act, wq, None, stride=stride, padding=k // 2)
    else:
        lo = F.interpolate(act, scale_factor=0.5, mode="nearest")  # back to the low-resolution operand
        ref = torch.zeros(batch, cout, 2 * h, 2 * w, dtype=torch.float64)
        pad = F.pad(lo, (1, 1, 1, 1))
        rows = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
        for py in (0, 1):
            for px in (0, 1):
                acc = 0
                for tr in (0, 1):
                    for tc in (0, 1):
                        wf = sum(wt[:, :, dy, dx] for dy in rows[py][tr] for dx in rows[px][tc])
                        wf = _rnd(wf, mode).double()
                        oy, ox = py + tr, px + tc     # offset into the padded low-resolution map
                        acc = acc + torch.einsum("oc,nchw->nohw", wf, pad[:, :, oy:oy + h, ox:ox + w])
                ref[:, :, py::2, px::2] = acc
    ref = ref + bias.double()[None, :, None, None]
    tproj = _t(7, (batch, cout + 5), 0.5)
    if temb:
        ref = ref + tproj[:, 3:3 + cout, None, None].double()
    r = _rnd(_t(8, tuple(ref.shape)), mode)
    if res:
        ref = ref + r.double()

    kind = ops.PACK_S2 if stride == 2 else (ops.PACK_FOLD if ups else ops.PACK_FWD)
    wpk = ops.pack_conv_weight(dv(wt), kind, mode)
    kw = dict(weight_h2=None, weight_h2_fold=None, weight_h2_s2=None)
    kw[{ops.PACK_FWD: "weight_h2", ops.PACK_FOLD: "weight_h2_fold", ops.PACK_S2: "weight_h2_s2"}[kind]] = wpk
    tp = dv(tproj)
    got, stats = ops.conv2d_fused(b0, ops.relayout_conv_weight(dv(wt)), dv(bias), src1=b1, ksize=k, stride=stride,
                                  upsample=ups, gn_scale_shift=ss, silu=gn, temb=tp[:, 3:] if temb else None,
                                  temb_stride=tp.stride(0), residual=_blk(r, mode) if res else None, cout=cout,
                                  src_blocked=True, dst_blocked=True, compute_dtype=mode, want_stats=True,
                                  weight_h2_stride=(cout + 63) // 64 * 64, **kw)
    assert got.dtype == TDT[mode]
    out = ops.from_blocked(got).cpu()
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) <= OUT_TOL[mode], rel_l2(out, ref)
    # element-wise: half an ulp of the stored result, plus noise that scales with the terms, not with the (possibly
    # cancelling) sum: operands whose fp32 activation sits on a rounding boundary may round the other way
    ulp = 2.0 ** -7 if mode == "bf16" else 2.0 ** -10   # spacing of the 16-bit type relative to the value (8 / 11-bit significands)
    rms = float(ref.pow(2).mean().sqrt())
    assert float(((out.double() - ref).abs() - 0.51 * ulp * ref.abs()).max()) <= 0.5 * ulp * rms
    if stats is not None:   # epilogue statistics describe the fp32 values before the rounding: within 2^-8 of the stored tensor's
        s_ref = ref.sum((2, 3))
        s_got = stats.cpu()[..., 0].sum(-1)
        assert float((s_got - s_ref).abs().max()) <= 2e-3 * float(ref.abs().sum((2, 3)).max())


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("lay", ["blk_to_nchw", "nchw_to_blk"])
def test_pointwise_mixed_layouts(mode, lay):
    """q/k/v projection (blocked 16-bit -> fp32 [N,3C,L], GroupNorm affine without SiLU) and out-projection
    (fp32 [N,C,L] -> blocked 16-bit + residual): the two pointwise layout pairs around the attention kernel."""
    n, c, h, w = 2, 64, 16, 32
    cout = 3 * c if lay == "blk_to_nchw" else c
    x = _t(11, (n, c, h, w))
    wt = _t(12, (cout, c, 1, 1), 1.0 / np.sqrt(c))
    bias = _t(13, (cout,), 0.1)
    wpk = ops.pack_conv_weight(wt.to(DEV), ops.PACK_FWD, mode)
    if lay == "blk_to_nchw":
        xq = _rnd(x, mode)
        gamma, beta = 1 + _t(14, (c,), 0.1), _t(15, (c,), 0.1)
        b0 = _blk(xq, mode)
        ss = ops.gn_scale_shift_from_parts(ops.gn_channel_stats_blocked(b0), gamma.to(DEV), beta.to(DEV), 32, 1e-5, h * w)
        act = xq.double() * ss.cpu()[:, :, 0, None, None].double() + ss.cpu()[:, :, 1, None, None].double()
        ref = F.conv2d(_rnd(act.float(), mode).double(), _rnd(wt, mode).double(), bias.double())
        got = ops.conv2d_fused(b0, ops.relayout_conv_weight(wt.to(DEV)), bias.to(DEV), ksize=1, gn_scale_shift=ss,
                               silu=False, cout=cout, src_blocked=True, dst_blocked=False, compute_dtype=mode,
                               weight_h2=wpk, weight_h2_stride=(cout + 63) // 64 * 64)
        assert got.dtype == torch.float32 and rel_l2(got.cpu(), ref) <= 2e-5
    else:
        r = _rnd(_t(16, (n, cout, h, w)), mode)
        ref = F.conv2d(_rnd(x, mode).double(), _rnd(wt, mode).double(), bias.double()) + r.double()
        got = ops.conv2d_fused(x.to(DEV), ops.relayout_conv_weight(wt.to(DEV)), bias.to(DEV), ksize=1, cout=cout,
                               residual=_blk(r, mode), src_blocked=False, dst_blocked=True, compute_dtype=mode,
                               weight_h2=wpk, weight_h2_stride=(cout + 63) // 64 * 64)
        assert got.dtype == TDT[mode] and rel_l2(ops.from_blocked(got).cpu(), ref) <= OUT_TOL[mode]


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("cin", [3, 4, 8])
def test_conv_in_fp32_image_to_blocked16(mode, cin):
    """conv_in: fp32 [N,C,H,W] image -> 16-bit channel-blocked activations on the exact fp32 MFMA chain."""
    x = _t(21, (2, cin, 32, 64))
    wt = _t(22, (64, cin, 3, 3), 1.0 / np.sqrt(cin * 9))
    bias = _t(23, (64,), 0.1)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    got = ops.conv2d_fused(x.to(DEV), ops.relayout_conv_weight(wt.to(DEV)), bias.to(DEV), cout=64, dst_blocked=True,
                           compute_dtype=mode)
    assert got.dtype == TDT[mode] and got.shape == (2, 8, 32, 64, 8)
    out = ops.from_blocked(got).cpu()
    assert torch.equal(out, _rnd(F.conv2d(x, wt, bias, padding=1), mode)) or rel_l2(out, ref) <= OUT_TOL[mode]


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("cout", [3, 4, 8])
def test_conv_out_blocked16_to_fp32_image(mode, cout):
    """conv_out: GroupNorm + SiLU folded in front, 16-bit blocked sources -> fp32 [N,C,H,W] on csrc/conv_out.hip: the
    activated input and the weights are rounded once to the 16-bit type, as torch.autocast does for this conv
    (tests/test_gpu_conv_out.py has the kernel's own cases; with tuning key 22 off: cout 8 on the zero-padded matrix-core
    kernel, cout <= 4 on the VALU kernel, which keeps the activated values in fp32)."""
    n, c, h, w = 2, 64, 32, 64
    xq = _rnd(_t(31, (n, c, h, w)), mode)
    wt = _t(32, (cout, c, 3, 3), 1.0 / np.sqrt(c * 9))
    bias = _t(33, (cout,), 0.1)
    gamma, beta = 1 + _t(34, (c,), 0.1), _t(35, (c,), 0.1)
    b0 = _blk(xq, mode)
    ss = ops.gn_scale_shift_from_parts(ops.gn_channel_stats_blocked(b0, splits=4), gamma.to(DEV), beta.to(DEV), 32, 1e-5, h * w)
    act = F.silu(xq.double() * ss.cpu()[:, :, 0, None, None].double() + ss.cpu()[:, :, 1, None, None].double())
    h2 = True
    ref = F.conv2d(_rnd(act.float(), mode).double() if h2 else act, _rnd(wt, mode).double() if h2 else wt.double(),
                   bias.double(), padding=1)
    kw = dict(weight_h2=ops.pack_conv_weight(wt.to(DEV), ops.PACK_FWD, mode), weight_h2_stride=64) if cout % 8 == 0 else {}
    got = ops.conv2d_fused(b0, ops.relayout_conv_weight(wt.to(DEV)), bias.to(DEV), gn_scale_shift=ss, silu=True, cout=cout,
                           src_blocked=True, dst_blocked=False, compute_dtype=mode, **kw)
    assert got.dtype == torch.float32 and got.shape == (n, cout, h, w)
    assert rel_l2(got.cpu(), ref) <= 2e-5, rel_l2(got.cpu(), ref)


def test_unsupported_mixed_shapes_fail_loudly():
    x = ops.to_blocked(_t(41, (1, 24, 16, 32)).to(DEV), "bf16")   # cin % 16 != 0: no 16-bit kernel takes it
    wt = _t(42, (64, 24, 3, 3)).to(DEV)
    with pytest.raises(RuntimeError, match="UNSUPPORTED_SHAPE"):
        ops.conv2d_fused(x, ops.relayout_conv_weight(wt), cout=64, src_blocked=True, dst_blocked=True, compute_dtype="bf16")
    with pytest.raises(RuntimeError, match="dtype"):
        ops.conv2d_fused(x.float(), ops.relayout_conv_weight(wt), cout=64, src_blocked=True, dst_blocked=True,
                         compute_dtype="bf16")


def test_layout_convert_and_stats_16bit():
    x = _t(51, (2, 16, 8, 32)) * 3
    for mode in ("bf16", "fp16"):
        b = ops.to_blocked(x.to(DEV), mode)
        assert b.dtype == TDT[mode] and b.shape == (2, 2, 8, 32, 8)
        assert torch.equal(b.cpu().permute(0, 1, 4, 2, 3).reshape(2, 16, 8, 32), x.to(TDT[mode]))
        assert torch.equal(ops.from_blocked(b).cpu(), _rnd(x, mode))
        st = ops.gn_channel_stats_blocked(b, splits=2).cpu().sum(2)
        xr = _rnd(x, mode).double()
        assert torch.allclose(st[..., 0], xr.sum((2, 3)), rtol=1e-12) and torch.allclose(st[..., 1], (xr * xr).sum((2, 3)), rtol=1e-12)


# ------------------------------------------------------------------------------------------------------------------
# whole network
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("cfg_name", ["CFG1", "CFG4_SMALL"])
def test_unet_forward_mixed_vs_fp32_oracle(mode, cfg_name):
    from oracle.unet_oracle import OracleUNet2DModel
    cfg = {"CFG1": CFG1, "CFG4_SMALL": CFG4_SMALL}[cfg_name]
    net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).eval().requires_grad_(False).set_compute_dtype(mode)
    ora = synth_weights(OracleUNet2DModel(**cfg)).eval()
    x = noisy_inputs(cfg, 2)
    t = torch.tensor([37, 801])
    with torch.no_grad():
        want = ora(x, t).sample
    got = net(x.to(DEV), t.to(DEV)).sample
    assert got.dtype == torch.float32 and torch.isfinite(got).all()
    e = rel_l2(got.cpu(), want)
    assert e <= 2e-2, e
    # and the fp32-equivalent engine on the same object after switching back (plan is rebuilt per dtype)
    net.set_compute_dtype("fp32")
    assert rel_l2(net(x.to(DEV), t.to(DEV)).sample.cpu(), want) <= 1e-4


def test_cfg5_default_net_bf16_256_vs_oracle():
    """BASELINE configs[4] network (256x256x8 raster, 56,580,360 parameters) forward in bf16 vs the fp32 oracle:
    rel-L2 <= 2e-2; batch rows independent of the batch they ride in."""
    net = synth_weights(d.UNet2DModel(**CFG5)).to(DEV).eval().requires_grad_(False).set_compute_dtype("bf16")
    assert sum(p.numel() for p in net.parameters()) == 56_580_360
    x = noisy_inputs(CFG5, 3)
    t = torch.tensor([980, 500, 20])
    got = net(x.to(DEV), t.to(DEV)).sample
    from tests.common import assert_matches_fullsize_golden, fullsize_case
    assert torch.equal(fullsize_case("cfg5_b3_row0_t980")[2], x[:1])
    assert_matches_fullsize_golden(got[:1], "cfg5_b3_row0_t980", rel=2e-2, ab=None)   # (the fp32 oracle's stored output)
    solo = net(x[1:2].to(DEV), t[1:2].to(DEV)).sample
    assert torch.equal(solo, got[1:2])


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("n,c,heads,l", [(2, 64, 8, 1024), (1, 512, 64, 256), (3, 32, 4, 96)])
def test_attention_single_product(mode, n, c, heads, l):
    """Self-attention core with ONE bf16 / fp16 matrix-core product per MAC (q, k, v and the probabilities rounded once,
    fp32 scores / max / denominators / accumulators) against fp64 softmax attention of the fp32 inputs: rounding-class
    error of the 16-bit type; every output row is still a convex combination of V rows (|out| <= max |v| over the rounded v)."""
    qkv = _t(61, (n, 3 * c, l), 1.3)
    d_head = c // heads
    q, k, v = (qkv[:, i * c:(i + 1) * c].double().view(n, heads, d_head, l) for i in range(3))
    att = torch.softmax(torch.einsum("nhdi,nhdj->nhij", q, k) / np.sqrt(d_head), -1)
    ref = torch.einsum("nhij,nhdj->nhdi", att, v).reshape(n, c, l)
    got = ops.attention(qkv.to(DEV), heads, dtype=mode).cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) <= (1.5e-2 if mode == "bf16" else 2e-3), rel_l2(got, ref)
    vmax = _rnd(qkv[:, 2 * c:], mode).abs().view(n, heads, d_head, l).amax(-1)      # per (n, head, dim)
    assert (got.view(n, heads, d_head, l).abs().amax(-1) <= vmax * (1 + 2e-2)).all()
    exact = ops.attention(qkv.to(DEV), heads).cpu()
    assert rel_l2(exact, ref) <= 1e-5
