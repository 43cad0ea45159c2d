"""Pre-staged operand images (dsg_conv_operand_prepare / dsg_conv_args.src_operand; csrc/conv_h2_kernel.h PRE form).

The fp32-equivalent 3x3 convs of the deep levels (the resnet convs of the network train.py:39-57 builds, evaluated at
training_pipeline.py:84 and inside DDPMPipeline.__call__) normalise, activate and fp16x2-split their halo patch in the K
loop -- once per cout tile, i.e. 4-8 times per patch at 256 / 512 output channels.  With an operand image that work is done
once by a streaming pass and the kernel DMAs its patches straight into LDS.  Checked here:
(a) the image itself against a torch evaluation of silu(x * scale + shift) and its fp16 pair;
(b) a conv with the image is BITWISE the same conv without it -- plain conv1 / conv2 (single and concatenated sources,
    with and without the residual), conv2 with the fused shortcut, the folded up-sampler conv with its range guard --
    including the GroupNorm statistics it leaves;
(c) the image path against fp64 to the split path's fp32-class bound;
(d) calls that cannot take an image refuse it; the query's pays-off rule;
(e) the whole network with and without it (dsg_set_tuning key 26): bitwise."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import _lib, ops, synth  # noqa: E402
from tests.common import CFG2, noisy_inputs, rel_l2, synth_weights  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def every_qualifying_conv_gets_an_image():
    """The launcher's default pays-off threshold (16 stagings per patch: the folded up-samplers) would leave the resnet convs
    on their own staging; the tests lower it to 4 (dsg_set_tuning key 27) so that every form of the PRE kernel runs."""
    lib = _lib.load()
    _lib.check(lib.dsg_set_tuning(27, 4))
    yield
    _lib.check(lib.dsg_set_tuning(27, 16))


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


def _ss(seed, n, c):
    return torch.stack([1 + _t(seed, (n, c), 0.2), _t(seed + 1, (n, c), 0.3)], -1).contiguous()


def test_operand_image_values():
    n, c0, c1, h, w = 2, 32, 16, 16, 32
    x0, x1, ss = _t(1, (n, c0, h, w)), _t(2, (n, c1, h, w)), _ss(3, n, c0 + c1)
    img = ops.conv_operand_prepare(ops.to_blocked(x0.to(DEV)), ops.to_blocked(x1.to(DEV)), ss.to(DEV), True).cpu()
    assert img.shape == (2, n, (c0 + c1) // 8, h + 2, w + 2, 8)
    # borders are the conv's zero padding
    for piece in range(2):
        assert (img[piece][:, :, 0] == 0).all() and (img[piece][:, :, -1] == 0).all()
        assert (img[piece][:, :, :, 0] == 0).all() and (img[piece][:, :, :, -1] == 0).all()
    hi = img[0][:, :, 1:-1, 1:-1].permute(0, 1, 4, 2, 3).reshape(n, c0 + c1, h, w).float()
    lo = img[1][:, :, 1:-1, 1:-1].permute(0, 1, 4, 2, 3).reshape(n, c0 + c1, h, w).float()
    x = torch.cat([x0, x1], 1).double()
    ref = F.silu(x * ss.double()[:, :, 0, None, None] + ss.double()[:, :, 1, None, None])
    got = hi.double() + lo.double() / 2048.0
    # the pair carries the fp32 value: 2^-22 relative (+ the fast exp / reciprocal of the activation: a few fp32 ulps)
    assert ((got - ref).abs() <= 2e-6 * ref.abs() + 1e-9).all()
    assert (hi == ref.float().half().float()).float().mean() > 0.99   # piece 0 is the value rounded to fp16


def _conv_pair(x0, x1, w, b, ss, temb, res, want_stats=True, **kw):
    """(with image, without) of one conv3x3(silu(norm(cat(x0, x1)))) call on channel-blocked fp32 tensors."""
    g = lambda t: None if t is None else t.to(DEV)
    x0b = ops.to_blocked(g(x0))
    x1b = ops.to_blocked(g(x1)) if x1 is not None else None
    resb = ops.to_blocked(g(res)) if res is not None else None
    cout = w.shape[0]
    common = dict(src1=x1b, ksize=3, cout=cout, gn_scale_shift=g(ss), silu=ss is not None, temb=g(temb), temb_stride=cout,
                  src_blocked=True, dst_blocked=True, weight_h2=ops.relayout_conv_weight_h2(g(w)), residual=resb,
                  want_stats=want_stats, **kw)
    outs = []
    for operand in ("auto", None):
        y = ops.conv2d_fused(x0b, None, g(b), operand=operand, **common)
        outs.append(y if want_stats else (y, None))
    return outs


# (n, c0, c1, cout, h, w, residual): the deep-level resnet convs at batch sizes that take the 16-row kernel
PLAIN = {
    "conv1_512_32": (16, 512, 0, 512, 32, 32, False),
    "conv1_cat_1024_512": (16, 512, 512, 512, 32, 32, False),
    "conv2_res_256_64": (8, 256, 0, 256, 64, 64, True),
    "conv1_cat_768_256": (8, 512, 256, 256, 64, 64, False),
}


@pytest.mark.parametrize("name", list(PLAIN))
def test_conv_with_operand_image_is_bitwise_the_staged_conv(name):
    n, c0, c1, cout, h, w, has_res = PLAIN[name]
    c = c0 + c1
    x0, x1 = _t(10, (n, c0, h, w)), (_t(11, (n, c1, h, w)) if c1 else None)
    wt, b = _t(12, (cout, c, 3, 3), 1.0 / np.sqrt(9 * c)), _t(13, (cout,), 0.1)
    ss, temb = _ss(14, n, c), _t(16, (n, cout), 0.3)
    res = _t(17, (n, cout, h, w)) if has_res else None
    g = lambda t: None if t is None else t.to(DEV)
    common = dict(src1=None if x1 is None else ops.to_blocked(g(x1)), ksize=3, cout=cout, gn_scale_shift=g(ss), silu=True,
                  src_blocked=True, dst_blocked=True, weight_h2=ops.relayout_conv_weight_h2(g(wt)))
    assert ops.conv2d_fused(ops.to_blocked(g(x0)), None, g(b), operand="query", **common), "the launcher should want an image here"
    (ya, sa), (yb, sb) = _conv_pair(x0, x1, wt, b, ss, temb, res)
    assert torch.equal(ya, yb), float((ya - yb).abs().max())
    assert sa is not None and torch.equal(sa, sb)
    # and both are the fp32-class evaluation of the reference expression
    x = (x0 if x1 is None else torch.cat([x0, x1], 1)).double()
    act = F.silu(x * ss.double()[:, :, 0, None, None] + ss.double()[:, :, 1, None, None])
    ref = F.conv2d(act, wt.double(), b.double(), padding=1) + temb.double()[:, :, None, None]
    if res is not None:
        ref = ref + res.double()
    scale = F.conv2d(act.abs(), wt.double().abs(), padding=1) + 1.0
    got = ops.from_blocked(ya).cpu().double()
    assert ((got - ref).abs() / scale).max().item() <= 6e-7
    assert rel_l2(got, ref) <= 2e-6


def test_fused_shortcut_conv_with_operand_image_is_bitwise():
    n, c, cout, sc0, sc1, h, w = 8, 256, 256, 256, 128, 64, 64
    hm, x0, x1 = _t(20, (n, c, h, w)), _t(21, (n, sc0, h, w), 40.0), _t(22, (n, sc1, h, w), 40.0)
    x0[1] *= 3e3   # (image 1 leaves the guard's safe range)
    x1[1] *= 3e3
    w2, wsc = _t(23, (cout, c, 3, 3), 1.0 / np.sqrt(9 * c)), _t(24, (cout, sc0 + sc1, 1, 1), 1.0 / np.sqrt(sc0 + sc1))
    b2, bsc, ss, temb = _t(25, (cout,), 0.1), _t(26, (cout,), 0.1), _ss(27, n, c), _t(29, (n, cout), 0.3)
    g = lambda t: t.to(DEV)
    x0b, x1b = ops.to_blocked(g(x0)), ops.to_blocked(g(x1))
    bound = ops.range_bound_from_stats(ops.gn_channel_stats_blocked(torch.cat([x0b, x1b], 1), splits=2))
    sc = dict(src0=x0b, src1=x1b, weight_h2=ops.relayout_conv_weight_h2(g(wsc)), bias=g(bsc), bound=bound)
    common = dict(ksize=3, cout=cout, gn_scale_shift=g(ss), silu=True, temb=g(temb), temb_stride=cout, src_blocked=True,
                  dst_blocked=True, weight_h2=ops.relayout_conv_weight_h2(g(w2)), want_stats=True, shortcut=sc)
    hb = ops.to_blocked(g(hm))
    ya, sa = ops.conv2d_fused(hb, None, g(b2), operand="auto", **common)
    yb, sb = ops.conv2d_fused(hb, None, g(b2), **common)
    assert torch.isfinite(ya).all()
    assert torch.equal(ya, yb) and torch.equal(sa, sb)


@pytest.mark.parametrize("mag", [1.0, 1e5, 1e-7])
def test_folded_upsampler_conv_with_operand_image_is_bitwise(mag):
    n, c, h, w = 16, 512, 32, 32
    x = _t(30, (n, c, h, w), mag)
    x[1] *= 30.0
    wt, b = _t(31, (c, c, 3, 3), 1.0 / np.sqrt(9 * c)), _t(32, (c,), 0.1)
    g = lambda t: t.to(DEV)
    xb = ops.to_blocked(g(x))
    bound = ops.range_bound_from_stats(ops.gn_channel_stats_blocked(xb, splits=2))
    common = dict(ksize=3, cout=c, upsample=True, src_blocked=True, dst_blocked=True, weight_h2=ops.relayout_conv_weight_h2(g(wt)),
                  weight_h2_fold=ops.pack_conv_weight(g(wt), kind=1), src_bound=bound, want_stats=True)
    assert ops.conv2d_fused(xb, None, g(b), operand="query", **common)
    ya, sa = ops.conv2d_fused(xb, None, g(b), operand="auto", **common)
    yb, sb = ops.conv2d_fused(xb, None, g(b), **common)
    assert torch.isfinite(ya).all()
    assert torch.equal(ya, yb) and torch.equal(sa, sb)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), wt.double(), b.double(), padding=1)
    for i in range(2):
        assert rel_l2(ops.from_blocked(ya)[i].cpu(), ref[i]) <= 3e-6


def test_calls_that_stage_their_own_patch_refuse_an_image():
    # a shallow-level conv (64 channels: one cout tile -- nothing to save) and a small batch (8-row tiles / split-K)
    g = lambda t: t.to(DEV)
    for n, c, cout, h, w in ((8, 64, 64, 128, 128), (1, 512, 512, 32, 32)):
        x, wt, ss = _t(40, (n, c, h, w)), _t(41, (cout, c, 3, 3), 0.05), _ss(42, n, c)
        xb = ops.to_blocked(g(x))
        common = dict(ksize=3, cout=cout, gn_scale_shift=g(ss), silu=True, src_blocked=True, dst_blocked=True,
                      weight_h2=ops.relayout_conv_weight_h2(g(wt)), splitk=True)
        assert not ops.conv2d_fused(xb, None, None, operand="query", **common)
    # ... and the library says so when handed one anyway
    img = ops.conv_operand_prepare(xb, None, g(ss), True)
    with pytest.raises(RuntimeError, match="stages its own patch"):
        ops.conv2d_fused(xb, None, None, operand=img, **common)


def test_whole_network_with_and_without_operand_images_is_bitwise():
    lib = _lib.load()
    net = synth_weights(d.UNet2DModel(**CFG2)).to(DEV).eval().requires_grad_(False)
    x = noisy_inputs(CFG2, 16).to(DEV)
    t = torch.full((16,), 500, dtype=torch.int64, device=DEV)
    y1 = net(x, t).sample.clone()
    _lib.check(lib.dsg_set_tuning(26, 0))
    try:
        y0 = net(x, t).sample.clone()
    finally:
        _lib.check(lib.dsg_set_tuning(26, 1))
    assert torch.isfinite(y1).all()
    assert torch.equal(y0, y1), float((y0 - y1).abs().max())
