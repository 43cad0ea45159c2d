"""A resnet's conv_shortcut fused into its conv2 (dsg_conv_args.sc_*, csrc/conv_h2_kernel.h SC form).

diffusers' ResnetBlock2D with in != out channels -- the first resnet of a wider down block and every up-block resnet of the
network train.py:39-57 builds (oracle/unet_oracle.py:57-63) -- computes
    output = conv_shortcut(input) + conv2(silu(norm2(h)))
The fused kernel contracts the 1x1 over the raw input as extra K-chunks on conv2's accumulators: no shortcut tensor in HBM,
no residual read.  Checked here: (a) against an fp64 evaluation of the same expression, to the split path's fp32-class
bound, for every kernel geometry that takes the fusion (16-row tiles, 8-row tiles, the one-weight-slab two-per-CU kernel),
single and concatenated shortcut sources; (b) against the unfused pair of calls (1x1, then 3x3 with residual) -- same
value up to the one extra rounding the unfused path has; (c) the range guard on the raw source at |x| ~ 1e5 / 1e-7 per
image; (d) batch independence: row i of a batch is bitwise the batch-1 call; (e) the GroupNorm statistics it leaves;
(f) the whole network with and without the fusion (dsg_set_tuning key 23)."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import _lib, ops, synth  # noqa: E402
from tests.common import CFG1, CFG2, noisy_inputs, rel_l2, same_kernels_at_any_batch, synth_weights  # noqa: E402

DEV = "cuda"


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


def _case(n, c, cout, sc0, sc1, h, w, mag=1.0, seed=0):
    """Tensors of one resnet tail: h (conv2's normalised source), the raw input (sc0 [+ sc1] channels), weights."""
    hm = _t(seed + 1, (n, c, h, w))
    x0 = _t(seed + 2, (n, sc0, h, w), mag)
    x1 = _t(seed + 3, (n, sc1, h, w), mag) if sc1 else None
    if n > 1:   # the range bound is per image
        x0[1] *= 30.0
        if x1 is not None:
            x1[1] *= 30.0
    w2 = _t(seed + 4, (cout, c, 3, 3), 1.0 / np.sqrt(9 * c))
    wsc = _t(seed + 5, (cout, sc0 + sc1, 1, 1), 1.0 / np.sqrt(sc0 + sc1))
    b2, bsc = _t(seed + 6, (cout,), 0.1), _t(seed + 7, (cout,), 0.1) * mag
    ss = torch.stack([1 + _t(seed + 8, (n, c), 0.2), _t(seed + 9, (n, c), 0.3)], -1).contiguous()
    tproj = _t(seed + 10, (n, cout), 0.3)
    return hm, x0, x1, w2, wsc, b2, bsc, ss, tproj


def _ref64(hm, x0, x1, w2, wsc, b2, bsc, ss, tproj):
    act = F.silu(hm.double() * ss.double()[:, :, 0, None, None] + ss.double()[:, :, 1, None, None])
    xin = x0.double() if x1 is None else torch.cat([x0.double(), x1.double()], 1)
    y = F.conv2d(act, w2.double(), b2.double(), padding=1) + tproj.double()[:, :, None, None]
    return y + F.conv2d(xin, wsc.double(), bsc.double())


def _bound_sum(hm, x0, x1, w2, wsc, ss):
    """sum |w||x| per output: the scale of the fp32-class error bound"""
    act = F.silu(hm.double() * ss.double()[:, :, 0, None, None] + ss.double()[:, :, 1, None, None]).abs()
    xin = (x0 if x1 is None else torch.cat([x0, x1], 1)).double().abs()
    return F.conv2d(act, w2.double().abs(), padding=1) + F.conv2d(xin, wsc.double().abs())


def _run(hm, x0, x1, w2, wsc, b2, bsc, ss, tproj, fused, guard=True, want_stats=False, splitk=False):
    g = lambda t: None if t is None else t.to(DEV)
    hb, x0b = ops.to_blocked(g(hm)), ops.to_blocked(g(x0))
    x1b = ops.to_blocked(g(x1)) if x1 is not None else None
    xcat = x0b if x1b is None else torch.cat([x0b, x1b], 1)
    bound = ops.range_bound_from_stats(ops.gn_channel_stats_blocked(xcat, splits=2)) if guard else None
    cout = w2.shape[0]
    wh2, whsc = ops.relayout_conv_weight_h2(g(w2)), ops.relayout_conv_weight_h2(g(wsc))
    common = dict(ksize=3, cout=cout, gn_scale_shift=g(ss), silu=True, temb=g(tproj), temb_stride=cout, src_blocked=True,
                  dst_blocked=True, weight_h2=wh2, want_stats=want_stats, splitk=splitk)
    if fused:
        sc = dict(src0=x0b, src1=x1b, weight_h2=whsc, bias=g(bsc), bound=bound)
        assert ops.conv2d_fused(hb, None, g(b2), shortcut=dict(sc, query_only=True), **common)
        out = ops.conv2d_fused(hb, None, g(b2), shortcut=sc, **common)
    else:
        r = ops.conv2d_fused(x0b, None, g(bsc), src1=x1b, ksize=1, cout=cout, src_blocked=True, dst_blocked=True,
                             weight_h2=whsc, src_bound=bound)
        out = ops.conv2d_fused(hb, None, g(b2), residual=r, **common)
    y, st = out if want_stats else (out, None)
    return ops.from_blocked(y).cpu(), st


# (n, c, cout, sc0, sc1, h, w): which kernel geometry serves conv2 --
#   deep16: 16-row tiles; rows8: 8-row tiles (grid too small for 16 rows); ws2: cin <= 128 on a >= 512-workgroup grid
SHAPES = {
    "deep16_cat": (8, 256, 256, 256, 128, 64, 64),
    "deep16_one": (8, 128, 256, 128, 0, 64, 64),
    "rows8_cat": (1, 256, 128, 192, 64, 32, 32),
    "ws2_cat": (8, 64, 64, 128, 64, 128, 128),
    "ws2_one": (8, 128, 128, 64, 0, 128, 64),
    "cout_tail": (2, 64, 96, 96, 32, 32, 64),     # cout not a multiple of the 64-channel tile
}


@pytest.mark.parametrize("name", list(SHAPES))
def test_fused_shortcut_matches_fp64_and_the_unfused_pair(name):
    n, c, cout, sc0, sc1, h, w = SHAPES[name]
    case = _case(n, c, cout, sc0, sc1, h, w, seed=zlib.crc32(name.encode()) % 1000)
    ref = _ref64(*case)
    scale = _bound_sum(case[0], case[1], case[2], case[3], case[4], case[7])
    fused, st = _run(*case, fused=True, want_stats=True)
    unfused, _ = _run(*case, fused=False)
    assert torch.isfinite(fused).all()
    # the yard-stick of SURVEY 8c: the same expression evaluated by torch-CPU in fp32
    hm, x0, x1, w2, wsc, b2, bsc, ss, tproj = case
    act = F.silu(hm * ss[:, :, 0, None, None] + ss[:, :, 1, None, None])
    cpu32 = (F.conv2d(act, w2, b2, padding=1) + tproj[:, :, None, None]) + F.conv2d(x0 if x1 is None else torch.cat([x0, x1], 1), wsc, bsc)
    for i in range(n):   # per image: image 1's shortcut term is 30x the conv2 term (the order of the two phases then shows)
        err = lambda y: ((y[i].double() - ref[i]).abs() / scale[i]).max().item()
        e_f, e_u, e_c = err(fused), err(unfused), err(cpu32)
        # fp32-class: within 2.5x of what torch's own fp32 evaluation loses (tiles that contract the shortcut FIRST add the
        # 3x3 products to an accumulator that already holds the larger sum: ~2x the rounding of the other order), and an
        # absolute 1e-6 of sum |w||x| (the separate calls hold 6e-7, tests/test_gpu_ops.py)
        assert e_f <= 2.5 * max(e_c, e_u) + 1e-7, (i, e_f, e_u, e_c)
        assert e_f <= 1e-6, (i, e_f)
    assert rel_l2(fused, ref) <= 2e-6 and rel_l2(fused, unfused) <= 2e-6
    # the statistics the next norm reads are those of the tensor written
    assert st is not None
    got_st, f64 = st.sum(2).cpu(), fused.double()
    assert ((got_st[..., 0] - f64.sum((2, 3))).abs() <= 3e-6 * f64.abs().sum((2, 3)) + 1e-4).all()   # (sums cancel: scale by sum |v|)
    assert ((got_st[..., 1] - (f64 ** 2).sum((2, 3))).abs() <= 3e-6 * (f64 ** 2).sum((2, 3)) + 1e-4).all()


@pytest.mark.parametrize("mag", [1e5, 1e-7])
@pytest.mark.parametrize("name", ["deep16_cat", "ws2_cat"])
def test_fused_shortcut_range_guard(name, mag):
    """The raw shortcut source at |x| ~ mag (image 1 another 30x): with the bound the fused kernel pre-scales the
    shortcut's patch by a power of two, brings conv2's partial sums to the same scale at the hand-over and scales
    everything back in the epilogue -- fp64 agreement as at magnitude 1; without the bound the fp16 pieces overflow."""
    n, c, cout, sc0, sc1, h, w = SHAPES[name]
    n = min(n, 2)
    case = _case(n, c, cout, sc0, sc1, h, w, mag=mag, seed=77)
    ref = _ref64(*case)
    got, _ = _run(*case, fused=True)
    assert torch.isfinite(got).all()
    for i in range(n):
        assert rel_l2(got[i], ref[i]) <= 3e-6, (i, rel_l2(got[i], ref[i]))
    if mag >= 1e5:
        assert not torch.isfinite(_run(*case, fused=True, guard=False)[0]).all()


def test_fused_shortcut_rows_do_not_depend_on_the_batch():
    """Row i of a batch-8 call == the batch-1 call on row i, bitwise -- although the two calls tile differently (batch 8:
    the two-workgroups-per-CU kernel; batch 1: 8-row tiles of the one-per-CU kernel) -- and the statistics agree bitwise too."""
    n, c, cout, sc0, sc1, h, w = SHAPES["ws2_cat"]
    case = _case(n, c, cout, sc0, sc1, h, w, seed=5)
    with same_kernels_at_any_batch():
        full, st = _run(*case, fused=True, want_stats=True)
        for i in (0, 1, 7):
            one = tuple(None if t is None else t[i:i + 1].contiguous() for t in (case[0], case[1], case[2])) + case[3:7] + (
                case[7][i:i + 1].contiguous(), case[8][i:i + 1].contiguous())
            got, st1 = _run(*one, fused=True, want_stats=True)
            assert torch.equal(got[0], full[i]), i
            assert torch.equal(st1[0], st[i]), i


@pytest.mark.parametrize("name,shape", [("deep_4slices", (1, 512, 512, 512, 512, 32, 32)), ("mid_2slices", (1, 256, 256, 512, 256, 64, 64)),
                                        ("b2_deep", (2, 512, 512, 512, 256, 32, 32))])
def test_fused_shortcut_under_split_k(name, shape):
    """Small batches (the reference samples at batch 1 and 5): a conv2 whose grid covers at most half the chip contracts its
    K in 2-4 parallel slices (32-cout workgroups where that still leaves CUs idle) -- each slice then takes its share of the
    shortcut's chunks as well, and the reduce pass adds the shortcut's bias.  Same value as the one-slice fused kernel to
    fp32 round-off, fp64 agreement unchanged, statistics from the reduce pass."""
    n, c, cout, sc0, sc1, h, w = shape
    case = _case(n, c, cout, sc0, sc1, h, w, seed=zlib.crc32(name.encode()) % 1000)
    ref = _ref64(*case)
    one, _ = _run(*case, fused=True)
    many, st = _run(*case, fused=True, want_stats=True, splitk=True)
    plain, _ = _run(*case, fused=False, splitk=True)
    assert torch.isfinite(many).all() and not torch.equal(one, many)       # (the sliced kernels really ran)
    assert rel_l2(many, ref) <= 2e-6 and rel_l2(many, one) <= 1e-6 and rel_l2(many, plain) <= 2e-6
    assert st is not None
    f64 = many.double()
    assert ((st.sum(2).cpu()[..., 0] - f64.sum((2, 3))).abs() <= 3e-6 * f64.abs().sum((2, 3)) + 1e-4).all()


@pytest.mark.parametrize("name,shape,fused", [("b5_deep_plain", (5, 512, 512, 512, 0, 32, 32), False),
                                              ("b5_deep_fused", (5, 512, 512, 512, 512, 32, 32), True),
                                              ("b1_bm64_4slices", (1, 512, 256, 512, 256, 64, 64), False),
                                              ("b1_bm64_4slices_fused", (1, 512, 256, 512, 256, 64, 64), True)])
def test_tile_height_rule_and_64_cout_slices_keep_the_values(name, shape, fused):
    """the selection rules of tuning key 36 are default-on and had no numeric test.  (i) batch-5 sampling at the
    32 x 32 level (generation.py:14-20): 129..170 eight-row tiles with long K take THREE K slices of 16-row tiles (NT = 4
    kernels, the fused-shortcut form included); (ii) batch 1 at the 64 x 64 level with >= 24 chunks of K: 64-cout workgroups with
    four slices instead of 32-cout workgroups with two.  Key 36 = 0 is the round-4 selection.  Both settings against fp64 in
    the split convs' round-off class and the statistics of what was written.  Rule (i) changes tile HEIGHT under the same
    three slices: bit-identical by construction (tile geometry never changes a bit); rule (ii) changes the number of K slices:
    another summation order, equal to fp32 round-off and really different bits."""
    n, c, cout, sc0, sc1, h, w = shape
    case = _case(n, c, cout, sc0, sc1, h, w, seed=zlib.crc32(name.encode()) % 1000)
    ref = _ref64(*case)
    scale = _bound_sum(case[0], case[1], case[2], case[3], case[4], case[7]) + 1e-30
    lib = _lib.load()
    got = {}
    try:
        for on in (0, 1):
            _lib.check(lib.dsg_set_tuning(36, on))
            got[on] = _run(*case, fused=fused, want_stats=True, splitk=True)
    finally:
        lib.dsg_set_tuning(36, 1)
    if n == 5:
        assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    else:
        assert not torch.equal(got[0][0], got[1][0]), "key 36 did not change the K slicing of this shape"
    for on in (0, 1):
        y, st = got[on]
        assert torch.isfinite(y).all()
        assert float(((y.double() - ref).abs() / scale).max()) <= 6e-7, on
        assert rel_l2(y, ref) <= 2e-6
        f64 = y.double()
        assert st is not None
        assert ((st.sum(2).cpu()[..., 0] - f64.sum((2, 3))).abs() <= 3e-6 * f64.abs().sum((2, 3)) + 1e-4).all()
    assert rel_l2(got[0][0], got[1][0]) <= 1e-6


def test_shapes_that_do_not_fuse_say_so():
    n, c, cout, sc0, sc1, h, w = 1, 64, 64, 64, 0, 16, 16      # a 16-wide map is narrower than a tile
    case = _case(n, c, cout, sc0, sc1, h, w)
    g = lambda t: t.to(DEV)
    hb, x0b = ops.to_blocked(g(case[0])), ops.to_blocked(g(case[1]))
    kw = dict(ksize=3, cout=cout, gn_scale_shift=g(case[7]), silu=True, src_blocked=True, dst_blocked=True,
              weight_h2=ops.relayout_conv_weight_h2(g(case[3])))
    sc = dict(src0=x0b, weight_h2=ops.relayout_conv_weight_h2(g(case[4])), bias=g(case[6]))
    assert not ops.conv2d_fused(hb, None, g(case[5]), shortcut=dict(sc, query_only=True), **kw)
    with pytest.raises(RuntimeError):
        ops.conv2d_fused(hb, None, g(case[5]), shortcut=sc, **kw)


@pytest.mark.parametrize("cfg_name,cfg,batch", [("cfg1", CFG1, 2), ("cfg2", CFG2, 2)])
def test_whole_net_with_and_without_the_fusion(cfg_name, cfg, batch):
    """dsg_unet_forward with the shortcuts fused (default) and as separate 1x1 calls (tuning key 23 = 0): the same eps to
    fp32 round-off; each is separately held to the oracle tolerance by the other suites."""
    lib = _lib.load()
    net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).eval().requires_grad_(False)
    x = noisy_inputs(cfg, batch).to(DEV)
    t = torch.tensor([980, 20][:batch], device=DEV)
    with same_kernels_at_any_batch():
        fused = net(x, t).sample.clone()
        try:
            _lib.check(lib.dsg_set_tuning(23, 0))
            plain = net(x, t).sample.clone()
        finally:
            lib.dsg_set_tuning(23, 1)
    assert torch.isfinite(fused).all()
    assert rel_l2(fused.cpu(), plain.cpu()) <= 5e-6, rel_l2(fused.cpu(), plain.cpu())
    if cfg_name == "cfg2":
        assert not torch.equal(fused, plain)   # (the fused kernels really ran: one rounding fewer per shortcut)


# 16-bit modes: (n, c, cout, sc0, sc1, h, w) -- 64-cout two-per-CU tiles, 128-cout tiles (16 and 8 rows), 8-row 64-cout tiles
SHAPES16 = {
    "bm64_16rows": (32, 64, 64, 128, 64, 64, 64),
    "bm128_16rows": (64, 128, 128, 128, 128, 32, 32),
    "bm128_8rows": (32, 64, 256, 64, 64, 8, 64),
    "bm64_8rows": (1, 128, 64, 128, 0, 32, 32),
}


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("name", list(SHAPES16))
def test_fused_shortcut_16bit_modes(name, mode):
    """bf16 / fp16 (train.py:24's mixed_precision; BASELINE configs[4]): the raw shortcut rows are 16-bit words that ARE the
    matrix-core operands -- one LDS-DMA per pixel row brings both channel blocks, no arithmetic.  Against fp64 on the
    rounded operands: the 16-bit rounding class, and closer than the two-call path, which rounds the shortcut's result to
    16 bits on its way through HBM."""
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[mode]
    rnd = lambda t: t.to(tdt).float()
    n, c, cout, sc0, sc1, h, w = SHAPES16[name]
    hm, x0, x1, w2, wsc, b2, bsc, ss, tproj = _case(n, c, cout, sc0, sc1, h, w, seed=zlib.crc32(name.encode()) % 1000)
    hm, x0 = rnd(hm), rnd(x0)
    x1 = rnd(x1) if x1 is not None else None
    # reference: operands rounded once (activation after the fp32 affine + SiLU; weights), fp64 accumulate
    act = rnd(F.silu(hm * ss[:, :, 0, None, None] + ss[:, :, 1, None, None])).double()
    xin = (x0 if x1 is None else torch.cat([x0, x1], 1)).double()
    ref = F.conv2d(act, rnd(w2).double(), b2.double(), padding=1) + tproj.double()[:, :, None, None] + \
        F.conv2d(xin, rnd(wsc).double(), bsc.double())
    g = lambda t: None if t is None else t.to(DEV)
    hb, x0b = ops.to_blocked(g(hm), mode), ops.to_blocked(g(x0), mode)
    x1b = ops.to_blocked(g(x1), mode) if x1 is not None else None
    cp = (cout + 63) // 64 * 64
    wh2, whsc = ops.pack_conv_weight(g(w2), ops.PACK_FWD, mode), ops.pack_conv_weight(g(wsc), ops.PACK_FWD, mode)
    common = dict(ksize=3, cout=cout, gn_scale_shift=g(ss), silu=True, temb=g(tproj), temb_stride=cout, src_blocked=True,
                  dst_blocked=True, weight_h2=wh2, weight_h2_stride=cp, compute_dtype=mode)
    sc = dict(src0=x0b, src1=x1b, weight_h2=whsc, bias=g(bsc))
    assert ops.conv2d_fused(hb, None, g(b2), shortcut=dict(sc, query_only=True), **common)
    fused = ops.from_blocked(ops.conv2d_fused(hb, None, g(b2), shortcut=sc, **common)).cpu()
    r = ops.conv2d_fused(x0b, None, g(bsc), src1=x1b, ksize=1, cout=cout, src_blocked=True, dst_blocked=True,
                         weight_h2=whsc, weight_h2_stride=cp, compute_dtype=mode)
    unfused = ops.from_blocked(ops.conv2d_fused(hb, None, g(b2), residual=r, **common)).cpu()
    assert torch.isfinite(fused).all()
    tol = {"bf16": 3e-3, "fp16": 8e-4}[mode]      # (tests/test_gpu_mixed.py: OUT_TOL)
    e_f, e_u = rel_l2(fused, ref), rel_l2(unfused, ref)
    assert e_f <= tol, e_f
    assert e_f <= e_u * 1.02 + 1e-6, (e_f, e_u)
    ulp = 2.0 ** -7 if mode == "bf16" else 2.0 ** -10
    rms = float(ref.pow(2).mean().sqrt())
    assert float(((fused.double() - ref).abs() - 0.51 * ulp * ref.abs()).max()) <= 0.5 * ulp * rms


def test_whole_net_bf16_with_and_without_the_fusion():
    from tests.common import CFG5
    lib = _lib.load()
    net = synth_weights(d.UNet2DModel(**CFG5)).to(DEV).eval().requires_grad_(False).set_compute_dtype("bf16")
    x = noisy_inputs(CFG5, 2).to(DEV)
    t = torch.tensor([980, 20], device=DEV)
    fused = net(x, t).sample.clone()
    try:
        _lib.check(lib.dsg_set_tuning(23, 0))
        plain = net(x, t).sample.clone()
    finally:
        lib.dsg_set_tuning(23, 1)
    assert torch.isfinite(fused).all() and not torch.equal(fused, plain)
    assert rel_l2(fused.cpu(), plain.cpu()) <= 1e-2   # two bf16 evaluations of the same net (each <= 2e-2 of the fp32 oracle)
