"""Parity of the backward / loss / optimizer kernels (through the C ABI) against torch-CPU autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from drivescenegen_amd import ops, synth  # noqa: E402
from tests.common import max_abs, rel_l2  # noqa: E402

DEV = "cuda"


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


def _close(got, want, rel=2e-5, ab=2e-5):
    got = got.cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all()
    assert rel_l2(got, want) <= rel, rel_l2(got, want)
    assert max_abs(got, want) <= ab * max(1.0, float(want.abs().max())), max_abs(got, want)


WG_CASES = [
    # name, c0, c1, cout, h, w, k, stride, ups, gn
    ("res64", 64, 0, 64, 16, 32, 3, 1, False, True),
    ("cat_straddle", 128, 64, 128, 8, 32, 3, 1, False, True),
    ("cin32_cit1", 32, 0, 64, 16, 32, 3, 1, False, True),
    ("h2_tiles_plain", 64, 0, 128, 32, 64, 3, 1, False, False),   # 2x2 column/row tiles, no norm: fp16x2-split wgrad
    ("h2_cat_even", 64, 64, 64, 8, 32, 3, 1, False, True),
    ("upsample", 64, 0, 64, 8, 16, 3, 1, True, False),
    ("stride2", 64, 0, 64, 16, 64, 3, 2, False, False),
    ("shortcut_1x1", 96, 32, 64, 8, 32, 1, 1, False, False),
    ("conv_in_c3", 3, 0, 32, 16, 32, 3, 1, False, False),
    ("conv_out_c3", 64, 0, 3, 16, 32, 3, 1, False, True),
    ("odd_direct", 16, 0, 8, 6, 10, 3, 1, False, True),
]


@pytest.mark.parametrize("case", WG_CASES, ids=[c[0] for c in WG_CASES])
def test_conv_backward(case):
    """wgrad (MFMA + VALU reference) and dgrad (forward kernel with transposed/flipped weights, zero-stuffed
    gather for stride 2, 2x2 sum-pool for the upsampler) against torch autograd."""
    name, c0, c1, cout, h, w, k, stride, ups, gn = case
    batch, cin = 3, c0 + c1
    x0 = _t(1, (batch, c0, h, w)).requires_grad_(True)
    x1 = _t(2, (batch, c1, h, w)).requires_grad_(True) if c1 else None
    wt = _t(3, (cout, cin, k, k), 1.0 / np.sqrt(cin * k * k)).requires_grad_(True)
    groups = 8 if cin % 8 == 0 else 1
    gamma, beta = 1 + _t(5, (cin,), 0.1), _t(6, (cin,), 0.1)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    a = F.silu(F.group_norm(xin, groups, gamma, beta, 1e-5)) if gn else xin
    a.retain_grad()
    u = F.interpolate(a, scale_factor=2.0, mode="nearest") if ups else a
    y = F.conv2d(u, wt, None, stride=stride, padding=k // 2)
    dy = _t(9, tuple(y.shape))
    y.backward(dy)

    d = lambda t: None if t is None else t.detach().to(DEV)
    ss = ops.gn_scale_shift(d(x0), d(gamma), d(beta), groups, 1e-5, src1=d(x1)) if gn else None
    for direct in (False, True):
        dw = torch.zeros_like(wt.detach()).to(DEV)
        ops.conv_wgrad(d(x0), d(dy), dw, src1=d(x1), ksize=k, stride=stride, upsample=ups, gn_scale_shift=ss,
                       silu=gn, direct=direct)
        _close(dw, wt.grad, rel=3e-5, ab=3e-5)
    # data gradient w.r.t. the conv input `a` (post-activation tensor)
    wd = ops.relayout_conv_weight_dgrad(d(wt))
    if stride == 2:
        da = ops.conv2d_fused(d(dy), wd, ksize=k, stride=1, upsample=2, cout=cin)
    elif ups:
        da = ops.conv2d_fused(d(dy), wd, ksize=k, stride=1, cout=cin, pool2=True)
    else:
        da = ops.conv2d_fused(d(dy), wd, ksize=k, stride=1, cout=cin)
    _close(da, a.grad, rel=3e-5, ab=3e-5)


@pytest.mark.parametrize("c0,c1,groups,hw,silu", [(64, 0, 32, (16, 16), True), (128, 64, 32, (8, 16), True),
                                                  (32, 0, 32, (8, 24), False), (64, 0, 32, (64, 64), True)])
def test_groupnorm_backward(c0, c1, groups, hw, silu):
    h, w = hw
    x0 = (_t(21, (2, c0, h, w)) * 2 + 0.7).requires_grad_(True)
    x1 = (_t(22, (2, c1, h, w)) - 0.3).requires_grad_(True) if c1 else None
    c = c0 + c1
    gamma = (1 + _t(23, (c,), 0.2)).requires_grad_(True)
    beta = _t(24, (c,), 0.2).requires_grad_(True)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    y = F.group_norm(xin, groups, gamma, beta, 1e-5)
    y = F.silu(y) if silu else y
    dy = _t(25, tuple(y.shape))
    y.backward(dy)
    d = lambda t: None if t is None else t.detach().to(DEV)
    ss, mr = ops.gn_scale_shift_train(d(x0), d(gamma), d(beta), groups, 1e-5, src1=d(x1))
    dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    add0 = _t(26, tuple(x0.shape))
    dx0, dx1 = ops.gn_bwd(d(x0), d(dy), ss, mr, d(gamma), groups, silu, dg, db, src1=d(x1), add0=d(add0))
    _close(dx0, x0.grad + add0, rel=2e-5, ab=2e-5)
    if c1:
        _close(dx1, x1.grad, rel=2e-5, ab=2e-5)
    _close(dg, gamma.grad, rel=2e-5, ab=1e-4)
    _close(db, beta.grad, rel=2e-5, ab=1e-4)


@pytest.mark.parametrize("n,c,heads,l", [(2, 64, 8, 1024), (1, 32, 4, 256), (1, 32, 2, 200)])
def test_attention_backward(n, c, heads, l):
    qkv = _t(31, (n, 3 * c, l), 1.2).requires_grad_(True)
    dd = c // heads
    q, k, v = [qkv[:, i * c:(i + 1) * c].view(n, heads, dd, l).transpose(2, 3) for i in range(3)]
    o = F.scaled_dot_product_attention(q, k, v).transpose(2, 3).reshape(n, c, l)
    do = _t(32, (n, c, l))
    o.backward(do)
    out, lse = ops.attention_train(qkv.detach().to(DEV), heads)
    _close(out, o.detach(), rel=1e-5, ab=1e-5)
    dqkv = ops.attention_bwd(qkv.detach().to(DEV), out, do.to(DEV), lse, heads)
    _close(dqkv, qkv.grad, rel=2e-5, ab=2e-5)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("n,c,heads,l", [(2, 64, 8, 1024), (3, 512, 64, 1024), (1, 32, 4, 256)])
def test_attention_backward_bf16_matrix_cores(n, c, heads, l, mode):
    """dsg_attention_bwd_dt(DSG_BF16 / DSG_F16): head_dim 8 on the matrix cores with q, k, v, dO, P, dS rounded once to 16 bits
    (the mixed-precision tapes; tiny gradients included: dO ~ 1e-6 as without a loss scale).  Against torch autograd in fp64."""
    qkv = _t(31, (n, 3 * c, l), 1.2).double().requires_grad_(True)
    dd = c // heads
    q, k, v = [qkv[:, i * c:(i + 1) * c].view(n, heads, dd, l).transpose(2, 3) for i in range(3)]
    o = F.scaled_dot_product_attention(q, k, v).transpose(2, 3).reshape(n, c, l)
    do = _t(32, (n, c, l)) * (1e-6 if l == 256 else 1.0)   # (one case at the magnitude of unscaled gradients)
    o.backward(do.double())
    out, lse = ops.attention_train(qkv.detach().float().to(DEV), heads, dtype=mode)   # (the mode's own forward: its lse)
    assert float((out.cpu().double() - o.detach()).norm() / o.detach().norm()) <= (1.2e-2 if mode == "bf16" else 2e-3)
    got = ops.attention_bwd(qkv.detach().float().to(DEV), out, do.to(DEV), lse, heads, dtype=mode).cpu().double()
    ref = qkv.grad
    assert torch.isfinite(got).all()
    for i, name in enumerate(("dq", "dk", "dv")):
        a, b = got[:, i * c:(i + 1) * c], ref[:, i * c:(i + 1) * c]
        err = float((a - b).norm() / b.norm())
        assert err <= (1.5e-2 if mode == "bf16" else 2.5e-3), (name, err)   # (fp16: dO scaled by a power of two, see attention.hip)
    out, lse = ops.attention_train(qkv.detach().float().to(DEV), heads)
    # the exact kernels through the same entry point (dtype 0) are unchanged
    exact = ops.attention_bwd(qkv.detach().float().to(DEV), out, do.to(DEV), lse, heads).cpu().double()
    assert float((exact - ref).norm() / ref.norm()) <= 1e-5


@pytest.mark.parametrize("n,c,heads,l,mag", [(2, 64, 8, 1024, 1.0), (3, 512, 64, 1024, 1.0), (1, 32, 4, 256, 1e-6), (2, 64, 8, 96, 30.0)])
def test_attention_backward_fp32_on_the_matrix_cores_is_in_the_valu_kernels_class(n, c, heads, l, mag):
    """Round 6: `dsg_attention_bwd` (the fp32 tape; training_pipeline.py:86 through the mid block's Attention) runs head_dim 8 on the
    matrix cores with every product as an fp16x2 split (three MFMAs, dO scaled to [1, 2) by a power of two first) instead of the
    two VALU kernels.  Against torch autograd in fp64, next to the VALU kernels (tuning key 38 = 0) on the same inputs: the same
    error class -- also for gradients of 1e-6 (no loss scale) and of 30 (a large one)."""
    from drivescenegen_amd import _lib
    qkv = _t(31, (n, 3 * c, l), 1.2).double().requires_grad_(True)
    dd = c // heads
    q, k, v = [qkv[:, i * c:(i + 1) * c].view(n, heads, dd, l).transpose(2, 3) for i in range(3)]
    o = F.scaled_dot_product_attention(q, k, v).transpose(2, 3).reshape(n, c, l)
    do = _t(32, (n, c, l)) * mag
    o.backward(do.double())
    ref = qkv.grad
    out, lse = ops.attention_train(qkv.detach().float().to(DEV), heads)
    lib = _lib.load()
    got = {}
    try:
        for on in (1, 0):
            _lib.check(lib.dsg_set_tuning(38, on))
            got[on] = ops.attention_bwd(qkv.detach().float().to(DEV), out, do.to(DEV), lse, heads).cpu().double()
    finally:
        lib.dsg_set_tuning(38, 1)
    assert not torch.equal(got[0], got[1])            # (another kernel really ran)
    for i, name in enumerate(("dq", "dk", "dv")):
        b = ref[:, i * c:(i + 1) * c]
        e1 = float((got[1][:, i * c:(i + 1) * c] - b).norm() / b.norm())
        e0 = float((got[0][:, i * c:(i + 1) * c] - b).norm() / b.norm())
        assert e1 <= max(2 * e0, 3e-6), (name, e1, e0)
        assert e1 <= 1e-5, (name, e1)
    assert torch.isfinite(got[1]).all()


def test_linear_silu_backward_and_sums():
    n, kf, mf = 5, 64, 96
    x = _t(41, (n, kf)).requires_grad_(True)
    w = _t(42, (mf, kf), 0.2).requires_grad_(True)
    b = _t(43, (mf,), 0.1).requires_grad_(True)
    z = F.linear(x, w, b)
    y = F.silu(z)
    dy = _t(44, (n, mf))
    y.backward(dy)
    dz = ops.silu_bwd(z.detach().to(DEV), dy.to(DEV))
    dw, db = torch.zeros(mf, kf, device=DEV), torch.zeros(mf, device=DEV)
    dx = ops.linear_bwd(x.detach().to(DEV), w.detach().to(DEV), dz, dw, db)
    _close(dx, x.grad)
    _close(dw, w.grad)
    _close(db, b.grad)
    t = _t(45, (3, 7, 9, 11))
    _close(ops.channel_sums(t.to(DEV)), t.sum((2, 3)), rel=1e-6, ab=1e-5)
    acc = torch.ones(7, device=DEV)
    _close(ops.reduce_rows_add(ops.channel_sums(t.to(DEV)), acc), 1 + t.sum((0, 2, 3)), rel=1e-6, ab=1e-5)


def test_mse_norm_adamw_match_torch():
    pred, tgt = _t(51, (4, 3, 32, 32)).requires_grad_(True), _t(52, (4, 3, 32, 32))
    loss = F.mse_loss(pred, tgt)
    loss.backward()
    gl, gd = ops.mse_loss(pred.detach().to(DEV), tgt.to(DEV))
    assert abs(float(gl.cpu()) - float(loss)) <= 1e-6 * float(loss)
    _close(gd, pred.grad, rel=1e-6, ab=1e-7)
    # clip_grad_norm_ + AdamW over 3 steps on a flat slab vs torch.optim.AdamW (train.py:66 defaults, lr 1e-5)
    p_ref = torch.nn.Parameter(_t(53, (5000,)))
    opt = torch.optim.AdamW([p_ref], lr=1e-3)
    p = p_ref.detach().clone().to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        g = _t(60 + step, (5000,), 3.0)
        p_ref.grad = g.clone()
        tn_ref = torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        opt.step()
        gg = g.to(DEV)
        tn = ops.l2_norm(gg)
        assert abs(float(tn.cpu()) - float(tn_ref)) <= 1e-5 * float(tn_ref)
        ops.adamw_step_(p, gg, m, v, step, lr=1e-3, total_norm=tn, max_norm=1.0)
        _close(p, p_ref.detach(), rel=1e-6, ab=1e-6)
    gg = _t(70, (5000,), 3.0).to(DEV)
    want = gg.cpu() * min(1.0, 1.0 / (float(gg.norm().cpu()) + 1e-6))
    _close(ops.clip_scale_(gg, ops.l2_norm(gg), 1.0), want, rel=1e-6, ab=1e-6)


def test_upsample_and_its_adjoint():
    """dsg_upsample_nearest2x == F.interpolate(nearest, x2); dsg_sumpool2x2 is its adjoint (+ add)."""
    x = _t(31, (2, 5, 6, 8))
    up = ops.upsample_nearest2x(x.to(DEV)).cpu()
    assert torch.equal(up, F.interpolate(x, scale_factor=2.0, mode="nearest"))
    g, a = _t(32, (2, 5, 12, 16)), _t(33, (2, 5, 6, 8))
    got = ops.sumpool2x2(g.to(DEV), add=a.to(DEV)).cpu()
    want = F.avg_pool2d(g.double(), 2) * 4 + a.double()
    assert torch.allclose(got.double(), want, rtol=0, atol=2e-6)
    assert torch.allclose(ops.sumpool2x2(g.to(DEV)).cpu().double(), F.avg_pool2d(g.double(), 2) * 4, rtol=0, atol=2e-6)


def _rand_wgrad_cases(n=8, seed=14555):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        c0 = int(rng.choice([32, 64, 96]))
        c1 = int(rng.choice([0, 0, 32, 64]))
        cout = int(rng.choice([64, 128, 192]))
        h, w = [(8, 32), (16, 32), (24, 64), (32, 32), (6, 32)][int(rng.integers(5))]
        out.append((c0, c1, cout, h, w, bool(rng.random() < 0.6), int(rng.integers(1, 4))))
    return out


@pytest.mark.parametrize("case", _rand_wgrad_cases(), ids=lambda c: "c%d+%d_o%d_%dx%d" % c[:5])
def test_wgrad_split_path_random_shapes(case):
    """3x3 weight gradient on the split path (row ring, split-K runs) vs torch autograd on seeded random shapes."""
    c0, c1, cout, h, w, gn, batch = case
    cin = c0 + c1
    x0 = _t(51, (batch, c0, h, w)).requires_grad_(True)
    x1 = _t(52, (batch, c1, h, w)).requires_grad_(True) if c1 else None
    wt = _t(53, (cout, cin, 3, 3), 1.0 / np.sqrt(cin * 9)).requires_grad_(True)
    gamma, beta = 1 + _t(55, (cin,), 0.1), _t(56, (cin,), 0.1)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    a = F.silu(F.group_norm(xin, 8, gamma, beta, 1e-5)) if gn else xin
    y = F.conv2d(a, wt, None, padding=1)
    dy = _t(59, tuple(y.shape))
    y.backward(dy)
    d = lambda t: None if t is None else t.detach().to(DEV)
    ss = ops.gn_scale_shift(d(x0), d(gamma), d(beta), 8, 1e-5, src1=d(x1)) if gn else None
    dw = torch.zeros_like(wt.detach()).to(DEV)
    # the split kernel also returns sum over pixels of dY per (n, cout) -- the bias / time-embedding gradients -- into a
    # strided table (column 0 is left alone); a shape it does not serve must refuse the request, not ignore it
    byp = ops.wgrad_h2_supported(c0, c1, cout, h, w)
    sums = torch.full((batch, cout + 1), 7.0, device=DEV)
    kw = dict(dy_sums=sums[:, 1:], dy_sums_stride=sums.stride(0))
    if byp:
        ops.conv_wgrad(d(x0), d(dy), dw, src1=d(x1), ksize=3, gn_scale_shift=ss, silu=gn, **kw)
        want = dy.double().sum((2, 3))
        assert torch.allclose(sums[:, 1:].cpu().double(), want, rtol=0, atol=2e-5 * float(want.abs().max()))
        assert bool((sums[:, 0] == 7.0).all())
    else:
        with pytest.raises(RuntimeError):
            ops.conv_wgrad(d(x0), d(dy), dw.clone(), src1=d(x1), ksize=3, gn_scale_shift=ss, silu=gn, **kw)
        ops.conv_wgrad(d(x0), d(dy), dw, src1=d(x1), ksize=3, gn_scale_shift=ss, silu=gn)
    _close(dw, wt.grad, rel=3e-5, ab=3e-5)


@pytest.mark.parametrize("case", [(64, 0, 128, 32, 64, True, 2), (128, 128, 256, 16, 32, True, 3), (96, 32, 384, 8, 32, False, 1),
                                  (256, 0, 512, 32, 32, True, 2), (64, 64, 128, 64, 64, True, 1)],
                         ids=lambda c: "c%d+%d_o%d_%dx%d" % c[:5])
def test_wgrad_wide_workgroups_match_the_32x64_ones(case):
    """cout % 128 == 0 selects the 32 ci x 128 co workgroup of the fp32-equivalent 3x3 weight gradient (conv_wgrad_h2w_kernel:
    two co tiles per wave, the centre tap's unit on pinned registers; dsg_set_tuning key 31 = 0 keeps the 32 x 64 workgroup).
    Both against fp64 autograd (3e-5 class, as the random-shape test) and against each other: per (ci, co, tap) the products are
    summed over a run's pixels in the same order, runs are cut by the (smaller) pair count, so the two agree to the rounding of
    the split-K reduce (1e-6), and the dY-sum by-products likewise."""
    from drivescenegen_amd import _lib
    c0, c1, cout, h, w, gn, batch = case
    cin = c0 + c1
    x0 = _t(71, (batch, c0, h, w))
    x1 = _t(72, (batch, c1, h, w)) if c1 else None
    gamma, beta = 1 + _t(75, (cin,), 0.1), _t(76, (cin,), 0.1)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    a = (F.silu(F.group_norm(xin.double(), 8, gamma.double(), beta.double(), 1e-5)) if gn else xin.double())
    dy = _t(79, (batch, cout, h, w))
    want = torch.nn.grad.conv2d_weight(a, (cout, cin, 3, 3), dy.double(), padding=1)
    d = lambda t: None if t is None else t.to(DEV)
    ss = ops.gn_scale_shift(d(x0), d(gamma), d(beta), 8, 1e-5, src1=d(x1)) if gn else None
    got = {}
    lib = _lib.load()
    try:
        for wide in (0, 1):
            _lib.check(lib.dsg_set_tuning(31, wide))
            dw = torch.zeros((cout, cin, 3, 3), device=DEV)
            sums = torch.zeros((batch, cout), device=DEV)
            ops.conv_wgrad(d(x0), d(dy), dw, src1=d(x1), ksize=3, gn_scale_shift=ss, silu=gn, dy_sums=sums, dy_sums_stride=cout)
            got[wide] = (dw.cpu(), sums.cpu())
    finally:
        lib.dsg_set_tuning(31, 1)
    for wide in (0, 1):
        _close(got[wide][0], want.float(), rel=3e-5, ab=3e-5)
    scale = float(want.abs().max())
    assert float((got[0][0].double() - got[1][0].double()).abs().max()) <= 2e-6 * scale
    assert torch.allclose(got[0][1], got[1][1], rtol=0, atol=2e-6 * float(dy.double().sum((2, 3)).abs().max()))


def test_wgrad_dy_sums_refused_where_not_a_by_product():
    """dy_sums is a by-product of the split 3x3 kernel (and of the 16-bit one); a call another kernel serves reports
    the request as unsupported instead of leaving the table unwritten."""
    x = _t(61, (2, 64, 8, 32)).to(DEV)
    dy = _t(62, (2, 64, 8, 32)).to(DEV)
    dw = torch.zeros(64, 64, 1, 1, device=DEV)
    sums = torch.zeros(2, 64, device=DEV)
    assert not ops.wgrad_h2_supported(64, 0, 64, 8, 32, ksize=1)
    with pytest.raises(RuntimeError):
        ops.conv_wgrad(x, dy, dw, ksize=1, dy_sums=sums, dy_sums_stride=sums.stride(0))
    ops.conv_wgrad(x, dy, dw, ksize=1)
    want = torch.einsum("nohw,nchw->oc", dy.double().cpu(), x.double().cpu())[:, :, None, None]
    _close(dw, want.float(), rel=3e-5, ab=3e-5)


PW_CASES = [
    # c0, c1, cout, h, w, batch, norm (GroupNorm affine on the input, no SiLU: the qkv projections), dY channel window
    (128, 0, 128, 8, 16, 3, False, None),      # 128 x 128 tiles, one run per stage pair
    (256, 128, 128, 16, 16, 2, False, None),   # concatenated sources, three ci blocks
    (96, 32, 64, 8, 32, 3, False, None),       # 64 x 64 tiles, sources meet inside a tile
    (128, 0, 128, 4, 32, 5, True, (384, 128)),  # normalised input, dY = channels [128, 256) of a fused [N, 384, L] tensor
    (192, 0, 64, 32, 32, 2, True, None),
]


@pytest.mark.parametrize("case", PW_CASES, ids=lambda c: "c%d+%d_o%d_%dx%d_b%d" % c[:6])
def test_wgrad_pointwise_split_path(case):
    """1x1 weight gradient on the fp16x2 split (pixel runs, 128- / 64-channel tiles) vs fp64 einsum."""
    c0, c1, cout, h, w, batch, norm, window = case
    cin = c0 + c1
    x0, x1 = _t(71, (batch, c0, h, w)), (_t(72, (batch, c1, h, w)) if c1 else None)
    ctot, coff = window if window else (cout, 0)
    dy = _t(73, (batch, ctot, h, w))
    xin = torch.cat([x0, x1], 1) if c1 else x0
    gamma, beta = 1 + _t(75, (cin,), 0.1), _t(76, (cin,), 0.1)
    a = F.group_norm(xin, 8, gamma, beta, 1e-5) if norm else xin
    want = torch.einsum("nohw,nchw->oc", dy[:, coff:coff + cout].double(), a.double())[:, :, None, None].float()
    d = lambda t: None if t is None else t.to(DEV)
    ss = ops.gn_scale_shift(d(x0), d(gamma), d(beta), 8, 1e-5, src1=d(x1)) if norm else None
    dw = torch.zeros(cout, cin, 1, 1, device=DEV)
    ops.conv_wgrad(d(x0), d(dy), dw, src1=d(x1), ksize=1, gn_scale_shift=ss, silu=False, cout=cout, dy_coff=coff)
    _close(dw, want, rel=3e-5, ab=3e-5)
    ops.conv_wgrad(d(x0), d(dy), dw, src1=d(x1), ksize=1, gn_scale_shift=ss, silu=False, cout=cout, dy_coff=coff)
    _close(dw, 2 * want, rel=3e-5, ab=6e-5)  # dw accumulates
