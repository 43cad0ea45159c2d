"""Range guard of the fp16x2-split path (csrc/conv_h2_kernel.h).  The split keeps fp32 accuracy
while operands sit in fp16's range; GroupNorm keeps most conv inputs there, but shortcut, up- / down-sampler convs and
the attention projections read the residual stream as it is.  Guard: (a) activations -- a per-image bound that rides on
the GroupNorm statistics (dsg_gn_finalize_parts_bound / dsg_range_bound_from_stats) lets those kernels pre-scale the patch
by an exact power of two; (b) weights -- max|w| is checked when a weight is uploaded and a conv whose weights leave
[2^-8, 3e4] runs on the exact f32 MFMA kernel.  No silent inf / nan, same tolerances as everywhere else."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import ops, synth  # noqa: E402
from oracle.unet_oracle import OracleUNet2DModel  # noqa: E402
from tests.common import CFG1, CFG4_SMALL, DEFAULT3, noisy_inputs, rel_l2, synth_weights  # noqa: E402

DEV = "cuda"


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


@pytest.mark.parametrize("mag", [1e5, 3e7, 1e-7, 1.0])
@pytest.mark.parametrize("kind", ["shortcut_1x1", "upsample_fold", "stride2", "plain3x3"])
def test_unnormalised_sources_at_any_magnitude(kind, mag):
    """A conv that reads its source without a norm, |x| ~ mag (image 1 another 100x larger: the bound is per image):
    with the bound from the tensor's statistics the split path matches fp64 to fp32 round-off; at mag = 1 the result
    is bit-identical to the unguarded call."""
    n, c, h, w, cout = 2, 64, 16, 64 if kind == "stride2" else 32, 64   # (stride 2: the result must still be a tile wide)
    x = _t(1, (n, c, h, w), mag)
    x[1] *= 100.0
    k, stride, ups = {"shortcut_1x1": (1, 1, False), "upsample_fold": (3, 1, True), "stride2": (3, 2, False),
                      "plain3x3": (3, 1, False)}[kind]
    wt = _t(2, (cout, c, k, k), 1.0 / np.sqrt(c * k * k))
    bias = _t(3, (cout,), 0.1) * mag
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(xin.double(), wt.double(), bias.double(), stride=stride, padding=k // 2)
    xb = ops.to_blocked(x.to(DEV))
    bound = ops.range_bound_from_stats(ops.gn_channel_stats_blocked(xb, splits=2))
    got_bound = bound.cpu().view(torch.float32)
    assert (got_bound >= x.abs().amax((1, 2, 3))).all() and (got_bound <= 64 * x.abs().amax((1, 2, 3))).all()
    kw = dict(weight_h2=ops.relayout_conv_weight_h2(wt.to(DEV)))
    if ups:
        kw = dict(weight_h2_fold=ops.relayout_conv_weight_h2_fold(wt.to(DEV)))
    if stride == 2:
        kw = dict(weight_h2_s2=ops.relayout_conv_weight_h2_s2(wt.to(DEV)))
    call = lambda b: ops.from_blocked(ops.conv2d_fused(xb, ops.relayout_conv_weight(wt.to(DEV)), bias.to(DEV), ksize=k,
                                                       stride=stride, upsample=ups, cout=cout, src_blocked=True,
                                                       dst_blocked=True, src_bound=b, **kw)).cpu()
    got = call(bound)
    assert torch.isfinite(got).all()
    for i in range(n):
        assert rel_l2(got[i], ref[i]) <= 2e-6, (i, rel_l2(got[i], ref[i]))
    if mag == 1.0:   # image 0 is inside the safe range: the guard multiplies by exactly 1
        assert torch.equal(got[0], call(None)[0])
    if mag >= 1e5:   # what the guard is for: without it the fp16 pieces overflow
        assert not torch.isfinite(call(None)).all()


def _with_scaled(cfg, scales):
    """engine + oracle with the named parameters multiplied by a factor"""
    net, ora = synth_weights(d.UNet2DModel(**cfg)), synth_weights(OracleUNet2DModel(**cfg)).eval()
    with torch.no_grad():
        for m in (net, ora):
            sd = dict(m.named_parameters())
            for name, f in scales.items():
                sd[name].mul_(f)
    return net.to(DEV).eval().requires_grad_(False), ora


@pytest.mark.parametrize("cfg_name", ["CFG1", "CFG4_SMALL"])
def test_residual_stream_grown_to_1e4_through_the_whole_net(cfg_name):
    """conv_in scaled so that the residual stream is ~1e4 (and one image another 30x): every shortcut / resampling conv
    reads it un-normalised.  Engine vs CPU oracle; both add O(1) resnet outputs to 1e4-sized values in fp32, which
    bounds the agreement at ~1e-3 of the small terms."""
    cfg = {"CFG1": CFG1, "CFG4_SMALL": CFG4_SMALL}[cfg_name]
    net, ora = _with_scaled(cfg, {"conv_in.weight": 1e4, "conv_in.bias": 1e4})
    x = noisy_inputs(cfg, 2)
    x[1] *= 30.0
    t = torch.tensor([980, 40])
    with torch.no_grad():
        want = ora(x, t).sample
    got = net(x.to(DEV), t.to(DEV)).sample.cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, want) <= 2e-3, rel_l2(got, want)


@pytest.mark.parametrize("factor", [1e-6, 1e6])
def test_weights_outside_the_split_range_take_the_exact_kernel(factor):
    """1e-6-scale (and 1e6-scale) weights in a resnet conv, a shortcut, a down- and an up-sampler conv: those convs are
    kept off the fp16x2 split (max|w| checked at upload), the network still matches the oracle at the fp32 tolerance."""
    names = ["down_blocks.0.resnets.1.conv2.weight", "down_blocks.1.resnets.0.conv_shortcut.weight",
             "down_blocks.0.downsamplers.0.conv.weight", "up_blocks.0.upsamplers.0.conv.weight",
             "mid_block.attentions.0.to_k.weight"]
    if factor > 1:   # (1e6 per layer compounds: four such layers in a row overflow torch-CPU's own fp32 GroupNorm variance, and
        names = [names[0], names[1], "mid_block.attentions.0.to_v.weight"]   # 1e12 scores its softmax -- the oracle must stay finite)
    net, ora = _with_scaled(CFG1, {n: factor for n in names})
    x = noisy_inputs(CFG1, 2)
    t = torch.tensor([500, 3])
    with torch.no_grad():
        want = ora(x, t).sample
    got = net(x.to(DEV), t.to(DEV)).sample.cpu()
    assert torch.isfinite(got).all() and rel_l2(got, want) <= 1e-4, rel_l2(got, want)


def test_default_net_results_unchanged_by_the_guard_in_the_normal_range():
    """The guard is exact and inactive for ordinary magnitudes: configs[1]-style forward still row-independent and
    within the fp32 tolerance of the oracle (the per-image bounds differ between rows)."""
    net = synth_weights(d.UNet2DModel(**DEFAULT3)).to(DEV).eval().requires_grad_(False)
    x = noisy_inputs(DEFAULT3, 3)
    t = torch.tensor([900, 450, 10])
    from tests.common import same_kernels_at_any_batch
    with same_kernels_at_any_batch():   # (batch 3 and batch 1 would cut K differently on the deep levels)
        got = net(x.to(DEV), t.to(DEV)).sample
        assert torch.equal(net(x[2:3].to(DEV), t[2:3].to(DEV)).sample, got[2:3])
    from tests.common import assert_matches_fullsize_golden, fullsize_case
    assert torch.equal(fullsize_case("default3_b3_row0_t900")[2], x[:1])
    assert_matches_fullsize_golden(got[:1], "default3_b3_row0_t900")   # (the oracle's stored output for this row)


@pytest.mark.parametrize("kind", ["deep3x3_gn", "pointwise", "stride2"])
def test_small_grid_split_k_matches_the_one_slice_kernel(kind):
    """Small batches (the reference samples at batch 1 and 5): a deep-level conv covers a fraction of the chip, so its K
    is contracted in parallel slices plus a reduce pass (dsg_conv_args.splitk_ws).  Same values as the one-slice kernel
    to fp32 round-off, identical statistics up to the summation order, fp64-reference accuracy unchanged."""
    n, h, w = 1, 32, 64 if kind == "stride2" else 32   # (stride 2: the result must still be a tile wide)
    c, cout, k, stride = {"deep3x3_gn": (512, 512, 3, 1), "pointwise": (1024, 512, 1, 1), "stride2": (256, 256, 3, 2)}[kind]
    x = _t(1, (n, c, h, w))
    wt = _t(2, (cout, c, k, k), 1.0 / np.sqrt(c * k * k))
    bias, r = _t(3, (cout,), 0.1), _t(4, (n, cout, h // stride, w // stride))
    tproj = _t(5, (n, cout), 0.3)
    xb = ops.to_blocked(x.to(DEV))
    ss, act = None, x.double()
    if kind == "deep3x3_gn":
        gamma, beta = (1 + _t(6, (c,), 0.1)), _t(7, (c,), 0.1)
        ss = ops.gn_scale_shift_from_parts(ops.gn_channel_stats_blocked(xb), gamma.to(DEV), beta.to(DEV), 32, 1e-5, h * w)
        act = F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5))
    ref = F.conv2d(act, wt.double(), bias.double(), stride=stride, padding=k // 2) + r.double()
    if kind == "deep3x3_gn":
        ref = ref + tproj.double()[:, :, None, None]
    kw = dict(weight_h2_s2=ops.relayout_conv_weight_h2_s2(wt.to(DEV))) if stride == 2 else dict(
        weight_h2=ops.relayout_conv_weight_h2(wt.to(DEV)))
    call = lambda split: ops.conv2d_fused(xb, ops.relayout_conv_weight(wt.to(DEV)), bias.to(DEV), ksize=k, stride=stride,
                                          gn_scale_shift=ss, silu=ss is not None, cout=cout,
                                          temb=tproj.to(DEV) if kind == "deep3x3_gn" else None, temb_stride=cout,
                                          residual=ops.to_blocked(r.to(DEV)), src_blocked=True, dst_blocked=True,
                                          want_stats=True, splitk=split, **kw)
    one, st_one = call(False)
    many, st_many = call(True)
    assert st_many is not None and st_many.shape[2] != st_one.shape[2]   # the split path really ran (its own tile count)
    a, b = ops.from_blocked(one).cpu(), ops.from_blocked(many).cpu()
    assert rel_l2(b, a) <= 1e-6 and rel_l2(b, ref) <= 2e-6 and rel_l2(a, ref) <= 2e-6
    assert torch.allclose(st_many.cpu().sum(2), st_one.cpu().sum(2), rtol=1e-5, atol=1e-3)


def test_tile_height_rule_keeps_the_bits_at_batch_5():
    """Batch-5 sampling (generation.py:14-20): the 64 x 64 level's convs are 160 tiles of 16 rows -- one round on 62 % of the CUs
    -- where the rule before the end of round 3 took 320 tiles of 8 rows in two rounds (dsg_set_tuning key 3 = 3 keeps that rule).
    The tile height moves pixels between workgroups, not products between sums: results and statistics are bit-identical."""
    from drivescenegen_amd import _lib
    n, c, cout, h, w = 5, 256, 256, 64, 64
    x, wt = _t(31, (n, c, h, w)), _t(32, (cout, c, 3, 3), 1.0 / np.sqrt(c * 9))
    bias, tproj = _t(33, (cout,), 0.1), _t(34, (n, cout), 0.3)
    xb = ops.to_blocked(x.to(DEV))
    gamma, beta = (1 + _t(35, (c,), 0.1)), _t(36, (c,), 0.1)
    ss = ops.gn_scale_shift_from_parts(ops.gn_channel_stats_blocked(xb), gamma.to(DEV), beta.to(DEV), 32, 1e-5, h * w)
    call = lambda: ops.conv2d_fused(xb, ops.relayout_conv_weight(wt.to(DEV)), bias.to(DEV), ksize=3, gn_scale_shift=ss, silu=True,
                                    cout=cout, temb=tproj.to(DEV), temb_stride=cout, src_blocked=True, dst_blocked=True,
                                    want_stats=True, weight_h2=ops.relayout_conv_weight_h2(wt.to(DEV)))
    lib = _lib.load()
    try:
        _lib.check(lib.dsg_set_tuning(3, 3))
        old, st_old = call()
        _lib.check(lib.dsg_set_tuning(3, 0))
        new, st_new = call()
    finally:
        lib.dsg_set_tuning(3, 0)
    assert torch.equal(old, new) and torch.equal(st_old, st_new)
    act = F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5))
    ref = F.conv2d(act, wt.double(), bias.double(), padding=1) + tproj.double()[:, :, None, None]
    assert rel_l2(ops.from_blocked(new).cpu(), ref) <= 2e-6
