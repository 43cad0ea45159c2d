"""Data gradient of the up-sampler conv (nearest-2x + 3x3, reference: diffusers Upsample2D inside the up blocks that
DriveSceneGen/utils/model/unet_2d.py builds) as ONE stride-2 conv with a 4x4 window over the full-resolution dY
(dsg_conv_args.s2_window4, dsg_conv_weight_pack kind 5).  Checked against torch-CPU fp64 autograd of the reference op on the
same (rounded) operands, and against the route it replaces (3x3 data gradient at full resolution, then 2x2 sums)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from drivescenegen_amd import ops, synth  # noqa: E402
from tests.common import rel_l2  # noqa: E402

DEV = "cuda"
TDT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
TOL = {"fp32": 2e-6, "bf16": 3e-3, "fp16": 8e-4}


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


def _reference(dy, w, h, wd):
    """fp64 autograd through conv(nearest2x(x), w): d/dx of <conv, dy>"""
    x = torch.zeros(dy.shape[0], w.shape[1], h, wd, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w.double(), None, padding=1)
    (gx,) = torch.autograd.grad(y, x, dy.double())
    return gx


CASES = [
    # name, conv cin (dX channels), conv cout (dY channels), low-res h, w, batch
    ("tile32_8rows", 64, 64, 8, 32, 2),
    ("cin128_cout64", 128, 64, 16, 32, 3),
    ("cin64_cout128_16rows", 64, 128, 32, 32, 16),    # 256 sixteen-row workgroups: the NT = 4 kernel
    ("narrow_16", 64, 32, 16, 16, 2),
    ("narrow_8", 32, 64, 8, 8, 2),
    ("cout_not_64", 40, 24, 8, 32, 1),                # padded cout tile, K = 4 * 24 = 96: six chunks
    ("wide_64cols", 32, 32, 8, 64, 1),
]


@pytest.mark.parametrize("mode", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_window4_matches_autograd(case, mode):
    name, cin, cout, h, w, n = case
    dy = _t(1, (n, cout, 2 * h, 2 * w)).to(TDT[mode]).float()
    wt = _t(2, (cout, cin, 3, 3), 1.0 / np.sqrt(9 * cout))
    add = _t(3, (n, cin, h, w)).to(TDT[mode]).float()
    pk = ops.pack_conv_weight(wt.to(DEV), ops.PACK_DGRAD_UPS, mode)
    # what the kernel multiplies: the window sums rounded ONCE to the operand type (fp32: the fp16 x 2 split, exact to 2^-22)
    ref = _reference(dy, wt, h, w)
    kw = dict(ksize=3, stride=2, cout=cin, src_blocked=True, dst_blocked=True, compute_dtype=mode, weight_h2_s2=pk, s2_window4=True)
    got = ops.from_blocked(ops.conv2d_fused(ops.to_blocked(dy.to(DEV), mode), None, **kw)).cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) <= TOL[mode], rel_l2(got, ref)
    # + residual (the gradient x already holds)
    got2 = ops.from_blocked(ops.conv2d_fused(ops.to_blocked(dy.to(DEV), mode), None, residual=ops.to_blocked(add.to(DEV), mode),
                                             **kw)).cpu()
    assert rel_l2(got2, ref + add.double()) <= TOL[mode]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_window4_fp32_nchw(case):
    """the fp32 tape's layout: [N,C,H,W] tensors on both sides (fp16 x 2 split products)"""
    name, cin, cout, h, w, n = case
    dy = _t(1, (n, cout, 2 * h, 2 * w))
    wt = _t(2, (cout, cin, 3, 3), 1.0 / np.sqrt(9 * cout))
    add = _t(3, (n, cin, h, w))
    ref = _reference(dy, wt, h, w)
    kw = dict(ksize=3, stride=2, cout=cin, weight_h2_s2=ops.pack_conv_weight(wt.to(DEV), ops.PACK_DGRAD_UPS, "fp32"), s2_window4=True)
    got = ops.conv2d_fused(dy.to(DEV), None, **kw).cpu()
    assert got.shape == (n, cin, h, w) and rel_l2(got, ref) <= 2e-6, rel_l2(got, ref)
    got2 = ops.conv2d_fused(dy.to(DEV), None, residual=add.to(DEV), **kw).cpu()
    assert rel_l2(got2, ref + add.double()) <= 2e-6
    blk = ops.from_blocked(ops.conv2d_fused(ops.to_blocked(dy.to(DEV), "fp32"), None, src_blocked=True, dst_blocked=True, **kw)).cpu()
    assert torch.equal(blk, got)   # same products in the same order as the channel-blocked form


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_window4_against_the_route_it_replaces(mode):
    """3x3 data gradient at full resolution + 2x2 sums: same value, two more roundings in the 16-bit modes (fp32: the tape's
    [N,C,H,W] tensors, bf16: its channel-blocked ones)"""
    n, cin, cout, h, w = 2, 64, 64, 16, 32
    dy = _t(5, (n, cout, 2 * h, 2 * w)).to(TDT[mode]).float().to(DEV)
    wt = _t(6, (cout, cin, 3, 3), 1.0 / np.sqrt(9 * cout)).to(DEV)
    blk = mode != "fp32"
    src = ops.to_blocked(dy, mode) if blk else dy
    lay = dict(src_blocked=blk, dst_blocked=blk, compute_dtype=mode)
    new = ops.conv2d_fused(src, None, ksize=3, stride=2, cout=cin, s2_window4=True,
                           weight_h2_s2=ops.pack_conv_weight(wt, ops.PACK_DGRAD_UPS, mode), **lay)
    full = ops.conv2d_fused(src, None, ksize=3, cout=cin, weight_h2=ops.pack_conv_weight(wt, ops.PACK_DGRAD, mode),
                            weight_h2_stride=(cin + 63) // 64 * 64, **lay)
    old = ops.sumpool2x2(full)
    if blk:
        new, old = ops.from_blocked(new), ops.from_blocked(old)
    assert rel_l2(new.cpu(), old.cpu()) <= (5e-6 if mode == "fp32" else 6e-3)


def test_window4_pack_is_the_window_sums():
    """kind 5 against a numpy restatement of its definition: element (k = (block, parity, j), tap, n)"""
    cout, cin = 16, 8
    wt = _t(7, (cout, cin, 3, 3))
    pk = ops.pack_conv_weight(wt.to(DEV), ops.PACK_DGRAD_UPS, "bf16").cpu().view(torch.bfloat16).float()
    img = pk.reshape(4 * cout // 16, 4, 2, 64, 8)   # [chunk][tap][g][n padded][j]
    rows = {0: (2,), 1: (1, 2), 2: (0, 1), 3: (0,)}
    for q in range(4 * cout // 16):
        for g in range(2):
            gi = 2 * q + g
            cb, py, px = gi >> 2, (gi >> 1) & 1, gi & 1
            for tap in range(4):
                i, j = 2 * (tap >> 1) - py + 1, 2 * (tap & 1) - px + 1
                want = sum(wt[cb * 8:(cb + 1) * 8, :, dy, dx] for dy in rows[i] for dx in rows[j])   # [8 (j)][cin]
                assert torch.equal(img[q, tap, g, :cin, :], want.to(torch.bfloat16).float().t())
    assert float(img[:, :, :, cin:, :].abs().max()) == 0.0


def test_window4_refused_where_the_kernel_does_not_serve():
    from drivescenegen_amd import _lib
    dy = torch.zeros(1, 8, 12, 64, 8, dtype=torch.bfloat16, device=DEV)   # 6 output rows: not a multiple of 8
    pk = ops.pack_conv_weight(torch.zeros(64, 64, 3, 3, device=DEV), ops.PACK_DGRAD_UPS, "bf16")
    with pytest.raises(_lib.DsgError):
        ops.conv2d_fused(dy, None, ksize=3, stride=2, cout=64, src_blocked=True, dst_blocked=True, compute_dtype="bf16",
                         weight_h2_s2=pk, s2_window4=True)
    x = torch.zeros(1, 64, 16, 64, device=DEV)   # [N,C,H,W] tensors with a 16-bit compute type: refused as well
    with pytest.raises(_lib.DsgError):
        ops.conv2d_fused(x, None, ksize=3, stride=2, cout=64, weight_h2_s2=pk, s2_window4=True, compute_dtype="bf16")


S2_CASES = [("tile32_16rows", 64, 64, 32, 64, 2), ("cin128_cout64_big", 128, 64, 64, 64, 16), ("narrow_16", 64, 128, 32, 32, 2),
            ("narrow_8", 32, 64, 16, 16, 3), ("cout_not_64", 24, 40, 16, 64, 1)]


@pytest.mark.parametrize("scale", [1.0, 3.0e5], ids=["unit", "range_guard"])
@pytest.mark.parametrize("case", S2_CASES, ids=[c[0] for c in S2_CASES])
def test_stride2_conv_fp32_nchw_on_the_space_to_depth_kernel(case, scale):
    """Downsample2D's conv (3x3, stride 2, padding 1; diffusers Downsample2D inside the down blocks of
    DriveSceneGen/utils/model/unet_2d.py's network) on fp32 [N,C,H,W] tensors -- the fp32 training tape's layout -- through the
    2x2-tap kernel over the space-to-depth image (dsg_set_tuning key 40; the channel-blocked form has served the inference plan
    since round 2): against F.conv2d in fp64, against the exact f32 kernel it replaces, statistics and range guard included
    (scale 3e5: the source is outside fp16's range and goes through the per-image power-of-two pre-scaling)."""
    from drivescenegen_amd import _lib
    name, cin, cout, h, w, n = case
    x = _t(11, (n, cin, h, w), scale)
    wt = _t(12, (cout, cin, 3, 3), 1.0 / np.sqrt(9 * cin))
    bias = _t(13, (cout,), 0.1)
    res = _t(14, (n, cout, h // 2, w // 2), scale)
    ref = F.conv2d(x.double(), wt.double(), bias.double(), stride=2, padding=1) + res.double()
    xd = x.to(DEV)
    bound = ops.range_bound_from_stats(ops.gn_channel_stats_blocked(ops.to_blocked(xd, "fp32")))
    kw = dict(ksize=3, stride=2, cout=cout, residual=res.to(DEV), weight_h2_s2=ops.pack_conv_weight(wt.to(DEV), ops.PACK_S2, "fp32"),
              want_stats=True, src_bound=bound)
    got, stats = ops.conv2d_fused(xd, None, bias.to(DEV), **kw)
    assert got.shape == (n, cout, h // 2, w // 2)
    assert rel_l2(got.cpu(), ref) <= 2e-6, rel_l2(got.cpu(), ref)
    assert stats is not None
    s_got = stats.cpu().sum(2)
    assert torch.allclose(s_got[..., 0], ref.sum((2, 3)), rtol=1e-5, atol=1e-4 * scale * scale)
    assert torch.allclose(s_got[..., 1], ref.pow(2).sum((2, 3)), rtol=1e-5)
    try:   # the exact kernel (needs the fp32 engine layout; no statistics from it)
        _lib.check(_lib.load().dsg_set_tuning(40, 0))
        with pytest.raises(RuntimeError):
            ops.conv2d_fused(xd, None, bias.to(DEV), **kw)
        old = ops.conv2d_fused(xd, ops.relayout_conv_weight(wt.to(DEV)), bias.to(DEV), ksize=3, stride=2, cout=cout, residual=res.to(DEV))
    finally:
        _lib.load().dsg_set_tuning(40, 1)
    assert rel_l2(got.cpu(), old.cpu()) <= 3e-6
