"""Per-op parity of the HIP kernels (through the C ABI) against plain torch-CPU fp32 references.

fp32 tolerances: contractions max|d| <= 2e-4*max(1,|y|inf) and rel-L2 <= 1e-4 (they are far
tighter in practice -- the f32 MFMA is an exact fmaf chain); scheduler math is bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from drivescenegen_amd import _lib, ops, synth  # noqa: E402
from oracle.scheduler_oracle import OracleDDIMScheduler, OracleDDPMScheduler  # noqa: E402
from oracle.unet_oracle import timestep_embedding  # noqa: E402
from tests.common import max_abs, rel_l2  # noqa: E402

DEV = "cuda"


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


def _check(got, want, tol_rel=1e-4, tol_abs=2e-4):
    got = got.cpu()
    assert got.shape == want.shape
    assert torch.isfinite(got).all()
    assert rel_l2(got, want) <= tol_rel, rel_l2(got, want)
    assert max_abs(got, want) <= tol_abs * max(1.0, float(want.abs().max())), max_abs(got, want)


CONV_CASES = [
    # name, c0, c1, cout, h, w, k, stride, ups, gn, silu, temb, res
    ("res_conv_64", 64, 0, 64, 32, 64, 3, 1, False, True, True, True, False),
    ("res_conv2_resid", 64, 0, 128, 16, 32, 3, 1, False, True, True, False, True),
    ("cout32_mt1", 32, 0, 32, 64, 64, 3, 1, False, True, True, True, True),
    ("concat_straddle", 128, 64, 128, 16, 32, 3, 1, False, True, True, True, False),
    ("upsample", 64, 0, 64, 16, 16, 3, 1, True, False, False, False, False),
    ("stride2", 64, 0, 64, 32, 64, 3, 2, False, False, False, False, False),
    ("stride2_narrow_out16", 64, 0, 64, 32, 32, 3, 2, False, False, False, False, True),   # 16-wide result: idle lanes
    ("narrow_16x16_f32", 24, 0, 32, 16, 16, 3, 1, False, True, True, True, True),
    ("shortcut_1x1_kc32", 128, 64, 64, 16, 32, 1, 1, False, False, False, False, False),
    ("proj_1x1_kc8", 40, 0, 64, 8, 32, 1, 1, False, True, False, False, True),
    ("conv_in_direct", 3, 0, 64, 32, 32, 3, 1, False, False, False, False, False),
    ("conv_out_direct", 64, 0, 4, 32, 32, 3, 1, False, True, True, False, False),
    ("odd_size_direct", 16, 0, 8, 13, 19, 3, 2, False, True, True, False, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize("batch", [1, 3])
def test_conv_fused(case, batch):
    name, c0, c1, cout, h, w, k, stride, ups, gn, silu, temb, res = case
    cin = c0 + c1
    x0 = _t(1, (batch, c0, h, w))
    x1 = _t(2, (batch, c1, h, w)) if c1 else None
    wt = _t(3, (cout, cin, k, k), 1.0 / np.sqrt(cin * k * k))
    bias = _t(4, (cout,), 0.1)
    groups = 8 if cin % 8 == 0 else 1
    gamma, beta = 1 + _t(5, (cin,), 0.1), _t(6, (cin,), 0.1)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    ref = xin
    if gn:
        ref = F.group_norm(ref, groups, gamma, beta, 1e-5)
        if silu:
            ref = F.silu(ref)
    if ups:
        ref = F.interpolate(ref, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(ref, wt, bias, stride=stride, padding=k // 2)
    tproj = _t(7, (batch, cout + 5), 0.5)
    if temb:
        ref = ref + tproj[:, 3:3 + cout, None, None]
    r = _t(8, tuple(ref.shape))
    if res:
        ref = ref + r

    d = lambda t: None if t is None else t.to(DEV)
    wr = ops.relayout_conv_weight(d(wt))
    ss = ops.gn_scale_shift(d(x0), d(gamma), d(beta), groups, 1e-5, src1=d(x1)) if gn else None
    tp = d(tproj)
    for direct in (False, True):
        got = ops.conv2d_fused(d(x0), wr, d(bias), src1=d(x1), ksize=k, stride=stride, upsample=ups,
                               gn_scale_shift=ss, silu=silu, temb=tp[:, 3:] if temb else None,
                               temb_stride=tp.stride(0), residual=d(r) if res else None, direct=direct,
                               cout=cout)
        _check(got, ref)


def test_conv_mfma_roundoff_class_vs_fp64():
    """v_mfma_f32_32x32x2_f32 is an exact-f32 fmaf chain in k order: against an fp64 reference its error
    stays in the f32 round-off class, <= 4e-7 * sum|a||b| at K = 2304 (the CPU path's blocked summation is
    a few times tighter; both are far inside the 2e-4 forward tolerance)."""
    x = _t(11, (2, 256, 32, 32))
    wt = _t(12, (64, 256, 3, 3), 1 / 48.0)
    ref64 = F.conv2d(x.double(), wt.double(), None, padding=1)
    mag = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1)
    got = ops.conv2d_fused(x.to(DEV), ops.relayout_conv_weight(wt.to(DEV))).cpu()
    assert float(((got.double() - ref64).abs() / mag).max()) <= 4e-7


@pytest.mark.parametrize("c0,c1,groups,hw", [(64, 0, 32, (32, 32)), (128, 64, 32, (16, 16)), (32, 0, 32, (8, 24)),
                                             (64, 0, 32, (256, 256))])
def test_groupnorm(c0, c1, groups, hw):
    h, w = hw
    x0 = _t(21, (2, c0, h, w)) * 2 + 0.7
    x1 = (_t(22, (2, c1, h, w)) - 0.3) if c1 else None
    c = c0 + c1
    gamma, beta = 1 + _t(23, (c,), 0.2), _t(24, (c,), 0.2)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    for silu in (False, True):
        ref = F.group_norm(xin, groups, gamma, beta, 1e-5)
        ref = F.silu(ref) if silu else ref
        ss = ops.gn_scale_shift(x0.to(DEV), gamma.to(DEV), beta.to(DEV), groups, 1e-5,
                                src1=None if x1 is None else x1.to(DEV))
        got = ops.gn_apply(xin.to(DEV), ss, silu)
        _check(got, ref, tol_rel=2e-6, tol_abs=5e-6)


@pytest.mark.parametrize("n,c,heads,l", [(2, 64, 8, 1024), (1, 512, 64, 1024), (3, 32, 4, 256), (1, 32, 2, 200),
                                         (1, 64, 2, 64),
                                         (1, 16, 2, 2048),   # head_dim 8: two LDS key tiles on the matrix-core kernel
                                         (2, 24, 3, 96),     # ... a partial key tile, a query block with idle waves
                                         (1, 8, 1, 100)])    # ... l % 32 != 0: the VALU kernel
def test_attention(n, c, heads, l):
    qkv = _t(31, (n, 3 * c, l), 1.5)
    d = c // heads
    q, k, v = [qkv[:, i * c:(i + 1) * c].view(n, heads, d, l).transpose(2, 3) for i in range(3)]
    ref = F.scaled_dot_product_attention(q, k, v)  # [n, heads, l, d]
    ref = ref.transpose(2, 3).reshape(n, c, l)
    got = ops.attention(qkv.to(DEV), heads)
    _check(got, ref, tol_rel=1e-5, tol_abs=1e-5)


@pytest.mark.parametrize("mode", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("n,c,heads,l", [(2, 512, 64, 1024), (1, 16, 2, 2048), (2, 24, 3, 96)])
def test_attention_blocked_layout(n, c, heads, l, mode):
    """dsg_attention_fwd_blocked: q, k, v and the output channel-blocked (head_dim 8 = one channel block), the layout the
    plan keeps between the q/k/v projection, the attention core and the out-projection.  fp32: the SAME arithmetic as the
    [N,3C,L] kernel, bit for bit; bf16 / fp16: q, k, v arrive rounded (the projection's store) instead of being rounded
    at the load -- compared with fp64 softmax attention of the rounded operands at the 16-bit rounding class."""
    qkv = _t(33, (n, 3 * c, l), 1.3)
    if mode == "fp32":
        want = ops.attention(qkv.to(DEV), heads)
        got = ops.from_blocked(ops.attention_blocked(ops.to_blocked(qkv.to(DEV)[:, :, :, None]), heads))
        assert torch.equal(got[:, :, :, 0], want)
        return
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[mode]
    qr = qkv.to(tdt).float()
    d_head = c // heads
    q, k, v = (qr[:, i * c:(i + 1) * c].double().view(n, heads, d_head, l) for i in range(3))
    att = torch.softmax(torch.einsum("nhdi,nhdj->nhij", q, k) / np.sqrt(d_head), -1)
    ref = torch.einsum("nhij,nhdj->nhdi", att, v).reshape(n, c, l)
    blk = ops.to_blocked(qkv.to(DEV)[:, :, :, None], mode)
    out = ops.attention_blocked(blk, heads, mode)
    assert out.dtype == tdt
    got = ops.from_blocked(out).cpu()[:, :, :, 0]
    assert torch.isfinite(got).all()
    from tests.common import rel_l2
    assert rel_l2(got, ref) <= (1.5e-2 if mode == "bf16" else 2e-3), rel_l2(got, ref)


def test_attention_spiked_scores():
    """Large score range: exercises the online-softmax rescale (one key dominates late in the sweep)."""
    n, c, heads, l = 1, 16, 2, 1024
    qkv = _t(32, (n, 3 * c, l))
    qkv[:, :c, 5] *= 6.0
    qkv[:, c:2 * c, 900] *= 8.0
    d = c // heads
    q, k, v = [qkv[:, i * c:(i + 1) * c].view(n, heads, d, l).transpose(2, 3).double() for i in range(3)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(2, 3).reshape(n, c, l)
    got = ops.attention(qkv.to(DEV), heads)
    _check(got, ref.float(), tol_rel=1e-5, tol_abs=1e-5)


def test_time_embedding_and_proj():
    ch, dim = 64, 256
    t = torch.tensor([0, 1, 499, 999, 37], dtype=torch.long)
    w1, b1 = _t(41, (dim, ch), 1 / 8.0), _t(42, (dim,), 0.1)
    w2, b2 = _t(43, (dim, dim), 1 / 16.0), _t(44, (dim,), 0.1)
    emb = timestep_embedding(t, ch)
    ref = F.silu(F.linear(F.silu(F.linear(emb, w1, b1)), w2, b2))
    got = ops.time_embed(t.to(DEV), w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV))
    _check(got, ref, tol_rel=2e-6, tol_abs=5e-6)
    wp, bp = _t(45, (777, dim), 1 / 16.0), _t(46, (777,), 0.1)
    _check(ops.linear(got, wp.to(DEV), bp.to(DEV)), F.linear(got.cpu(), wp, bp), tol_rel=2e-6, tol_abs=5e-6)


def test_add_noise_bit_exact():
    import drivescenegen_amd as d
    o, s = OracleDDPMScheduler(), d.DDPMScheduler()
    assert torch.equal(o.alphas_cumprod, s.alphas_cumprod)
    x0, nz = _t(51, (4, 3, 64, 64)).clamp(-1, 1), _t(52, (4, 3, 64, 64))
    t = torch.tensor([0, 3, 500, 999])
    got = s.add_noise(x0.to(DEV), nz.to(DEV), t.to(DEV)).cpu()
    assert torch.equal(got, o.add_noise(x0, nz, t))
    # train.py:91 form: a single HWC image with timesteps=[100]
    img = x0[0].permute(1, 2, 0).contiguous()
    got = s.add_noise(img.to(DEV), nz[0].permute(1, 2, 0).contiguous().to(DEV), torch.LongTensor([100])).cpu()
    assert torch.equal(got, o.add_noise(img, nz[0].permute(1, 2, 0).contiguous(), torch.LongTensor([100])))


@pytest.mark.parametrize("steps", [10, 50, 750, 1000])
def test_ddpm_step_bit_exact(steps):
    import drivescenegen_amd as d
    o, s = OracleDDPMScheduler(), d.DDPMScheduler()
    o.set_timesteps(steps)
    s.set_timesteps(steps)
    assert torch.equal(o.timesteps, s.timesteps) and s.timesteps.dtype == torch.int64
    x, e, z = _t(61, (2, 4, 32, 32)) * 1.3, _t(62, (2, 4, 32, 32)), _t(63, (2, 4, 32, 32))
    for t in [int(s.timesteps[0]), int(s.timesteps[len(s.timesteps) // 2]), int(s.timesteps[-2]), 0]:
        want = o.step(e, t, x, noise=z).prev_sample
        got = s.step(e.to(DEV), t, x.to(DEV), variance_noise=z.to(DEV)).prev_sample.cpu()
        assert torch.equal(got, want), (steps, t, max_abs(got, want))


@pytest.mark.parametrize("steps", [10, 50, 100])
def test_ddim_step_bit_exact(steps):
    import drivescenegen_amd as d
    o, s = OracleDDIMScheduler(), d.DDIMScheduler()
    o.set_timesteps(steps)
    s.set_timesteps(steps)
    assert torch.equal(o.timesteps, s.timesteps)
    x, e = _t(71, (2, 4, 32, 32)) * 1.3, _t(72, (2, 4, 32, 32))
    for t in [int(v) for v in s.timesteps[[0, len(s.timesteps) // 2, -1]]]:
        want = o.step(e, t, x).prev_sample
        got = s.step(e.to(DEV), t, x.to(DEV)).prev_sample.cpu()
        assert torch.equal(got, want), (steps, t, max_abs(got, want))


def test_postprocess_modes():
    x = _t(81, (2, 3, 16, 16)) * 1.2
    ref = (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1)
    assert torch.equal(ops.postprocess(x.to(DEV), 0).cpu(), ref)
    assert np.array_equal(ops.postprocess(x.to(DEV), 1).cpu().numpy(), (ref.numpy() * 255).round().astype("uint8"))
    assert np.array_equal(ops.postprocess(x.to(DEV), 2).cpu().numpy(), (ref * 255.).numpy().astype(np.uint8))


def test_errors_are_reported_not_fatal():
    from drivescenegen_amd import _lib
    x = torch.zeros(1, 8, 16, 16, device=DEV)
    w = torch.zeros(8, 9, 8, device=DEV)
    with pytest.raises(_lib.DsgError, match="stride"):
        ops.conv2d_fused(x, w, stride=3)
    with pytest.raises(_lib.DsgError, match="head_dim"):
        ops.attention(torch.zeros(1, 3 * 24, 16, device=DEV), 2)


H2_CASES = [
    # name, c0, c1, cout, h, w, ups, gn, temb, res
    ("h2_res64", 64, 0, 64, 32, 64, False, True, True, False),
    ("h2_resid_128", 64, 0, 128, 16, 32, False, True, False, True),
    ("h2_concat_straddle", 128, 64, 128, 16, 32, False, True, True, False),
    ("h2_upsample", 64, 0, 64, 16, 16, True, False, False, False),
    ("h2_cout96_pad", 32, 0, 96, 8, 32, False, True, False, True),
    ("h2_narrow_16x16", 64, 0, 64, 16, 16, False, True, True, True),       # map narrower than a tile: masked lanes
    ("h2_narrow_8x8_concat", 32, 32, 128, 8, 8, False, True, False, False),
    ("h2_upsample_fold", 64, 0, 128, 16, 32, True, False, False, False, 3, True),   # x2 folded into 2x2 phase kernels
    ("h2_upsample_fold_rows8", 32, 0, 64, 8, 32, True, False, False, True, 3, True),
    # pointwise (shortcut / attention projections): k = 1 flagged by a trailing element
    ("h2_1x1_shortcut_concat", 128, 64, 128, 16, 32, False, False, False, False, 1),
    ("h2_1x1_gn_res_16x16", 64, 0, 192, 16, 16, False, True, False, True, 1),
    ("h2_1x1_rows16", 32, 0, 64, 32, 32, False, False, False, True, 1),
]


def test_layout_convert_round_trip_and_blocked_stats():
    """[N,C,H,W] <-> [N,C/8,H,W,8] (dsg_layout_convert) and the GroupNorm statistics of a blocked tensor."""
    x = _t(11, (3, 24, 5, 7), 2.0).to(DEV)
    xb = ops.to_blocked(x)
    want = x.view(3, 3, 8, 5, 7).permute(0, 1, 3, 4, 2).contiguous()
    assert torch.equal(xb, want)
    assert torch.equal(ops.from_blocked(xb), x)
    ref = torch.stack([x.double().sum((2, 3)), (x.double() ** 2).sum((2, 3))], -1).cpu()
    for splits in (1, 5, 7):
        st = ops.gn_channel_stats_blocked(xb, splits).cpu()
        assert st.shape == (3, 24, splits, 2)
        assert torch.allclose(st.sum(2), ref, rtol=1e-12, atol=1e-12)
    part = x.double()[:, :, :1, :5].reshape(3, 24, -1)  # the first of 7 splits = the first 5 pixels
    assert torch.allclose(ops.gn_channel_stats_blocked(xb, 7)[:, :, 0, 0].cpu(), part.sum(-1).cpu(), rtol=1e-12, atol=1e-12)


F32_BLOCKED_CASES = [
    # name, c0, c1, cout, h, w, k, stride, ups, gn, src_blocked, dst_blocked
    ("conv_in_4ch_to_blocked", 4, 0, 64, 16, 32, 3, 1, False, False, False, True),
    ("downsample_s2_blocked", 32, 0, 32, 16, 64, 3, 2, False, False, True, True),
    ("f32_3x3_concat_gn_res_blocked", 24, 8, 40, 8, 32, 3, 1, False, True, True, True),
    ("f32_gather_x2_blocked", 16, 0, 32, 8, 16, 3, 1, True, False, True, True),
    ("f32_1x1_blocked_to_plain", 48, 0, 96, 8, 32, 1, 1, False, True, True, False),
    ("conv_out_fewout_from_blocked", 64, 0, 4, 16, 32, 3, 1, False, True, True, False),
    # the layers of the tiny (32, 64)-channel net that the split path does not take (cout % 64 != 0)
    ("tiny_conv_in_3_to_32", 3, 0, 32, 64, 64, 3, 1, False, False, False, True),
    ("tiny_res32_gn_temb", 32, 0, 32, 64, 64, 3, 1, False, True, True, True, True),
    ("tiny_up_96_to_32_concat_temb", 64, 32, 32, 64, 64, 3, 1, False, True, True, True, True),
    ("tiny_up_96_to_32_shortcut_1x1", 64, 32, 32, 64, 64, 1, 1, False, False, True, True),
    ("tiny_conv_out_32_to_3", 32, 0, 3, 64, 64, 3, 1, False, True, True, False),
]


@pytest.mark.parametrize("case", F32_BLOCKED_CASES, ids=[c[0] for c in F32_BLOCKED_CASES])
def test_conv_f32_kernels_take_blocked_layouts(case):
    """The f32 matrix-core kernel (conv_in, stride-2 down-samplers, fallback shapes) and the conv_out kernel with
    channel-blocked sources / results: bit-identical to their [N,C,H,W] calls."""
    name, c0, c1, cout, h, w, k, stride, ups, gn, sb, db = case[:12]
    temb = len(case) > 12 and case[12]
    batch, cin = 2, c0 + c1
    d = lambda t: None if t is None else t.to(DEV)
    x0, x1 = d(_t(1, (batch, c0, h, w), 1.7)), (d(_t(2, (batch, c1, h, w))) if c1 else None)
    wt = d(_t(3, (cout, cin, k, k), 1.0 / np.sqrt(cin * k * k)))
    bias = d(_t(4, (cout,), 0.1))
    gamma, beta = d(1 + _t(5, (cin,), 0.1)), d(_t(6, (cin,), 0.1))
    hc, wc = (2 * h, 2 * w) if ups else (h, w)
    ho, wo = (hc + 2 * (k // 2) - k) // stride + 1, (wc + 2 * (k // 2) - k) // stride + 1
    res = name.startswith("f32_3x3")
    r = d(_t(8, (batch, cout, ho, wo))) if res else None
    wr = ops.relayout_conv_weight(wt)  # (cout is zero-padded to a multiple of 32 columns)
    ss = ops.gn_scale_shift(x0, gamma, beta, 8, 1e-5, src1=x1) if gn else None
    tp = d(_t(7, (batch, cout + 5), 0.5))
    kw = dict(ksize=k, stride=stride, upsample=ups, gn_scale_shift=ss, silu=gn, cout=cout,
              temb=tp[:, 3:] if temb else None, temb_stride=tp.stride(0))
    want = ops.conv2d_fused(x0, wr, bias, src1=x1, residual=r, **kw)
    b = lambda t: None if t is None else ops.to_blocked(t)
    run = lambda: ops.conv2d_fused(b(x0) if sb else x0, wr, bias, src1=b(x1) if sb else x1, residual=b(r) if db else r,
                                   src_blocked=sb, dst_blocked=db, **kw)
    if "conv_in" in name or "conv_out" in name:
        # image -> blocked and blocked -> image calls now have kernels of their own (csrc/conv_in.hip, conv_out.hip;
        # tests/test_gpu_conv_in.py, test_gpu_conv_out.py): same values to round-off; with them switched off (tuning keys
        # 21 / 22) the f32 / VALU kernels serve the calls as before, bit for bit
        near = run()
        near = ops.from_blocked(near) if db else near
        assert float((near - want).abs().max()) <= 3e-6 * float(want.abs().max())
        lib = _lib.load()
        _lib.check(lib.dsg_set_tuning(21, 0))
        _lib.check(lib.dsg_set_tuning(22, 0))
        try:
            got = run()
        finally:
            _lib.check(lib.dsg_set_tuning(21, 1))
            _lib.check(lib.dsg_set_tuning(22, 1))
    else:
        got = run()
    got = ops.from_blocked(got) if db else got
    assert torch.equal(got, want), (name, float((got - want).abs().max()))


@pytest.mark.parametrize("case", [c for c in H2_CASES if (len(c) <= 10 or c[10] == 3) and not c[6] and c[1] + c[2] <= 128
                                  and c[5] % 32 == 0], ids=lambda c: c[0])
def test_conv_h2_small_workgroup_geometry_is_bit_identical(case):
    """The shallow levels' geometry (32 couts x 8 rows per workgroup, two workgroups per CU; dsg_set_tuning key 16):
    the same contraction order per output, so results and epilogue statistics equal the 64 x 16 geometry's bit for
    bit."""
    from drivescenegen_amd import _lib
    name, c0, c1, cout, h, w, ups, gn, temb, res = case[:10]
    batch, cin = 2, c0 + c1
    d = lambda t: None if t is None else t.to(DEV)
    b = lambda t: None if t is None else ops.to_blocked(t)
    x0, x1 = d(_t(1, (batch, c0, h, w), 1.7)), (d(_t(2, (batch, c1, h, w))) if c1 else None)
    wt = d(_t(3, (cout, cin, 3, 3), 1.0 / np.sqrt(cin * 9)))
    bias = d(_t(4, (cout,), 0.1))
    gamma, beta = d(1 + _t(5, (cin,), 0.1)), d(_t(6, (cin,), 0.1))
    tp = d(_t(7, (batch, cout + 5), 0.5))
    r = d(_t(8, (batch, cout, h, w))) if res else None
    wr, wh = ops.relayout_conv_weight(wt), ops.relayout_conv_weight_h2(wt)
    ss = ops.gn_scale_shift(x0, gamma, beta, 8, 1e-5, src1=x1) if gn else None
    kw = dict(src1=b(x1), ksize=3, gn_scale_shift=ss, silu=gn, temb=tp[:, 3:] if temb else None,
              temb_stride=tp.stride(0), residual=b(r), cout=cout, weight_h2=wh, src_blocked=True, dst_blocked=True,
              want_stats=True)
    lib = _lib.load()
    try:
        _lib.check(lib.dsg_set_tuning(16, 0))
        want, wstats = ops.conv2d_fused(b(x0), wr, bias, **kw)
        _lib.check(lib.dsg_set_tuning(16, 2))  # (2: take the small geometry whatever the grid size)
        got, gstats = ops.conv2d_fused(b(x0), wr, bias, **kw)
    finally:
        lib.dsg_set_tuning(16, 0)
    assert torch.equal(got, want), float((got - want).abs().max())
    assert torch.equal(gstats, wstats)


@pytest.mark.parametrize("c,cout,h,w", [(32, 64, 16, 64), (64, 128, 32, 64), (8, 72, 16, 64)])
def test_conv_h2_stride2_space_to_depth(c, cout, h, w):
    """Downsample2D's stride-2 3x3 conv on the split path: a 2x2 conv over the space-to-depth image, addressed
    straight out of the channel-blocked tensor (dsg_conv_args.weight_h2_s2).  Against torch in fp64, with the round-off
    class of the other split convs; the f32-MFMA kernel on the same call as the yardstick; epilogue statistics."""
    batch = 2
    d = lambda t: t.to(DEV)
    x = _t(41, (batch, c, h, w), 1.3)
    wt = _t(42, (cout, c, 3, 3), 1.0 / np.sqrt(9 * c))
    bias = _t(43, (cout,), 0.1)
    ref64 = F.conv2d(x.double(), wt.double(), bias.double(), stride=2, padding=1)
    mag = F.conv2d(x.double().abs(), wt.double().abs(), None, stride=2, padding=1) + 1e-30
    wr, ws2 = ops.relayout_conv_weight(d(wt)), ops.relayout_conv_weight_h2_s2(d(wt))
    xb = ops.to_blocked(d(x))
    kw = dict(ksize=3, stride=2, cout=cout, src_blocked=True, dst_blocked=True)
    got32 = ops.from_blocked(ops.conv2d_fused(xb, wr, d(bias), **kw)).cpu()
    got, stats = ops.conv2d_fused(xb, wr, d(bias), weight_h2_s2=ws2, want_stats=True, **kw)
    got = ops.from_blocked(got).cpu()
    assert not torch.equal(got, got32)  # (another kernel really ran)
    e32 = float(((got32.double() - ref64).abs() / mag).max())
    eh2 = float(((got.double() - ref64).abs() / mag).max())
    assert eh2 <= max(2 * e32, 3e-7), (eh2, e32)
    assert stats is not None and stats.shape[:2] == (batch, cout)
    want_st = torch.stack([got.double().sum((2, 3)), (got.double() ** 2).sum((2, 3))], -1)
    assert torch.allclose(stats.sum(2).cpu(), want_st, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mode,c,cout,h,w,batch", [("splitk", 512, 512, 16, 16, 8), ("splitk", 256, 128, 8, 8, 3), ("fold", 512, 512, 16, 16, 2),
                                                   ("fold", 64, 72, 8, 8, 1), ("s2", 256, 256, 32, 32, 2), ("s2", 64, 128, 16, 16, 3)])
def test_narrow_maps_take_splitk_fold_and_stride2_kernels(mode, c, cout, h, w, batch):
    """Maps narrower than the 32-column tile (BASELINE configs[3]: 512 channels at 16 x 16, the up-sampler out of it and the
    down-sampler into it) on the kernels the wide maps use -- split-K slices + reduce pass, the folded up-sampler, the stride-2
    space-to-depth kernel (dsg_set_tuning key 32; before round 4: one-slice / exact f32-MFMA kernels).  Against fp64 in the
    split convs' round-off class, against the key-32-off call of the same arguments (fp32 round-off: another summation order),
    and the epilogue statistics of what was written."""
    from drivescenegen_amd import _lib
    d = lambda t: None if t is None else t.to(DEV)
    x = _t(61, (batch, c, h, w), 1.2)
    wt = _t(62, (cout, c, 3, 3), 1.0 / np.sqrt(9 * c))
    bias = _t(63, (cout,), 0.1)
    gamma, beta = 1 + _t(64, (c,), 0.1), _t(65, (c,), 0.1)
    kw = dict(ksize=3, cout=cout, src_blocked=True, dst_blocked=True, want_stats=True)
    xb = ops.to_blocked(d(x))
    wr = ops.relayout_conv_weight(d(wt))
    if mode == "splitk":    # a resnet conv: GroupNorm + SiLU in front, bias + residual behind
        r = _t(66, (batch, cout, h, w))
        a = F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5))
        want = F.conv2d(a, wt.double(), bias.double(), padding=1) + r.double()
        mag = F.conv2d(a.abs(), wt.double().abs(), None, padding=1) + 1e-30
        ss = ops.gn_scale_shift(d(x), d(gamma), d(beta), 32, 1e-5)
        kw.update(gn_scale_shift=ss, silu=True, residual=ops.to_blocked(d(r)), weight_h2=ops.relayout_conv_weight_h2(d(wt)), splitk=True)
    elif mode == "fold":
        a = F.interpolate(x.double(), scale_factor=2.0, mode="nearest")
        want = F.conv2d(a, wt.double(), bias.double(), padding=1)
        mag = F.conv2d(a.abs(), wt.double().abs(), None, padding=1) + 1e-30
        kw.update(upsample=True, weight_h2_fold=ops.relayout_conv_weight_h2_fold(d(wt)))
    else:
        want = F.conv2d(x.double(), wt.double(), bias.double(), stride=2, padding=1)
        mag = F.conv2d(x.double().abs(), wt.double().abs(), None, stride=2, padding=1) + 1e-30
        kw.update(stride=2, weight_h2_s2=ops.relayout_conv_weight_h2_s2(d(wt)))
    lib = _lib.load()
    got = {}
    try:
        for on in (0, 1):
            _lib.check(lib.dsg_set_tuning(32, on))
            y, st = ops.conv2d_fused(xb, wr, d(bias), **kw)
            got[on] = (ops.from_blocked(y).cpu(), st)
    finally:
        lib.dsg_set_tuning(32, 1)
    assert not torch.equal(got[0][0], got[1][0])   # (another kernel really ran)
    for on in (0, 1):
        err = float(((got[on][0].double() - want).abs() / mag).max())
        assert err <= 6e-7, (on, err)
    assert float((got[0][0].double() - got[1][0].double()).abs().max()) <= 4e-6 * float(want.abs().max())
    y, st = got[1]
    assert st is not None
    ref = torch.stack([y.double().sum((2, 3)), (y.double() ** 2).sum((2, 3))], -1)
    assert torch.allclose(st.sum(2).cpu(), ref, rtol=3e-6, atol=1e-4)


@pytest.mark.parametrize("case", H2_CASES, ids=[c[0] for c in H2_CASES])
def test_conv_h2_blocked_layout_is_bit_identical(case):
    """Channel-blocked activations (dsg_conv_args.src_layout / dst_layout = 1): the same arithmetic in the same
    order as the [N,C,H,W] call of the split path, so the results are equal bit for bit -- every supported pair of
    layouts (3x3: all tensors blocked; pointwise: any pair)."""
    name, c0, c1, cout, h, w, ups, gn, temb, res = case[:10]
    batch, cin, k = 2, c0 + c1, (case[10] if len(case) > 10 else 3)
    fold = len(case) > 11 and case[11]
    if (ups and not fold) or cout % 8 or c0 % 8 or c1 % 8:
        pytest.skip("no blocked variant of this call")
    d = lambda t: None if t is None else t.to(DEV)
    x0, x1 = d(_t(1, (batch, c0, h, w), 1.7)), (d(_t(2, (batch, c1, h, w))) if c1 else None)
    wt = d(_t(3, (cout, cin, k, k), 1.0 / np.sqrt(cin * k * k)))
    bias = d(_t(4, (cout,), 0.1))
    gamma, beta = d(1 + _t(5, (cin,), 0.1)), d(_t(6, (cin,), 0.1))
    ho, wo = (2 * h, 2 * w) if ups else (h, w)
    tp = d(_t(7, (batch, cout + 5), 0.5))
    r = d(_t(8, (batch, cout, ho, wo))) if res else None
    wr, wh = ops.relayout_conv_weight(wt), ops.relayout_conv_weight_h2(wt)
    whf = ops.relayout_conv_weight_h2_fold(wt) if fold else None
    ss = ops.gn_scale_shift(x0, gamma, beta, 8, 1e-5, src1=x1) if gn else None
    kw = dict(ksize=k, upsample=ups, gn_scale_shift=ss, silu=gn, temb=tp[:, 3:] if temb else None,
              temb_stride=tp.stride(0), cout=cout, weight_h2=wh, weight_h2_fold=whf)
    want, wstats = ops.conv2d_fused(x0, wr, bias, src1=x1, residual=r, want_stats=True, **kw)
    b = lambda t: None if t is None else ops.to_blocked(t)
    pairs = [(True, True)] + ([(True, False), (False, True)] if k == 1 else [])
    for sb, db in pairs:
        for stats in (True, False):  # (the epilogue has a code path of its own for each)
            got = ops.conv2d_fused(b(x0) if sb else x0, wr, bias, src1=b(x1) if sb else x1,
                                   residual=b(r) if db else r, src_blocked=sb, dst_blocked=db, want_stats=stats, **kw)
            got, gstats = got if stats else (got, None)
            got = ops.from_blocked(got) if db else got
            assert torch.equal(got, want), (name, sb, db, stats, float((got - want).abs().max()))
            if stats:
                assert (wstats is None) == (gstats is None)
                if wstats is not None and gstats.shape == wstats.shape:
                    assert torch.equal(gstats, wstats)


@pytest.mark.parametrize("case", H2_CASES, ids=[c[0] for c in H2_CASES])
def test_conv_h2_split_matches_fp32(case):
    """fp16x2-split matrix-core path (conv_h2.hip): same contract as the fp32 kernel, fp32-class accuracy."""
    name, c0, c1, cout, h, w, ups, gn, temb, res = case[:10]
    batch, cin, k = 2, c0 + c1, (case[10] if len(case) > 10 else 3)
    x0 = _t(1, (batch, c0, h, w), 1.7)
    x1 = _t(2, (batch, c1, h, w)) if c1 else None
    wt = _t(3, (cout, cin, k, k), 1.0 / np.sqrt(cin * k * k))
    bias = _t(4, (cout,), 0.1)
    gamma, beta = 1 + _t(5, (cin,), 0.1), _t(6, (cin,), 0.1)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    ref = F.silu(F.group_norm(xin, 8, gamma, beta, 1e-5)) if gn else xin
    ref = F.interpolate(ref, scale_factor=2.0, mode="nearest") if ups else ref
    ref64 = F.conv2d(ref.double(), wt.double(), bias.double(), padding=k // 2)
    mag = F.conv2d(ref.double().abs(), wt.double().abs(), None, padding=k // 2) + 1e-30
    tproj = _t(7, (batch, cout + 5), 0.5)
    r = _t(8, tuple(ref64.shape))
    extra = torch.zeros_like(ref64)
    if temb:
        extra = extra + tproj[:, 3:3 + cout, None, None].double()
    if res:
        extra = extra + r.double()
    d = lambda t: None if t is None else t.to(DEV)
    wr, wh = ops.relayout_conv_weight(d(wt)), ops.relayout_conv_weight_h2(d(wt))
    fold = len(case) > 11 and case[11]
    whf = ops.relayout_conv_weight_h2_fold(d(wt)) if fold else None
    ss = ops.gn_scale_shift(d(x0), d(gamma), d(beta), 8, 1e-5, src1=d(x1)) if gn else None
    tp = d(tproj)
    kw = dict(src1=d(x1), ksize=k, upsample=ups, gn_scale_shift=ss, silu=gn, temb=tp[:, 3:] if temb else None,
              temb_stride=tp.stride(0), residual=d(r) if res else None, cout=cout)
    got32 = ops.conv2d_fused(d(x0), wr, d(bias), **kw).cpu()
    goth2 = ops.conv2d_fused(d(x0), wr, d(bias), weight_h2=wh, weight_h2_fold=whf, **kw).cpu()
    if fold:  # really another code path than the gathered up-sampler conv, same result to fp32 round-off
        gather = ops.conv2d_fused(d(x0), wr, d(bias), weight_h2=wh, **kw).cpu()
        assert not torch.equal(gather, goth2) and (gather - goth2).abs().max() <= 1e-5 * max(1.0, float(gather.abs().max()))
    _check(goth2, (ref64 + extra).float())
    e32 = float(((got32.double() - ref64 - extra).abs() / mag).max())
    eh2 = float(((goth2.double() - ref64 - extra).abs() / mag).max())
    # round-off class of the contraction: the split path is not worse than the fp32 chain (plus the shared
    # staging error of the fast SiLU)
    assert eh2 <= max(2 * e32, 3e-7), (eh2, e32)
    assert not torch.equal(got32, goth2)  # the two kernels really are different code paths


STATS_CASES = [
    # name, c0, c1, cout, h, w, k, ups
    ("stats_3x3_rows8", 32, 0, 64, 16, 32, 3, False),
    ("stats_3x3_rows16_concat", 64, 32, 128, 32, 64, 3, False),
    ("stats_1x1_16x16", 64, 0, 128, 16, 16, 1, False),
    ("stats_upsample", 64, 0, 64, 16, 16, 3, True),
    ("stats_upsample_fold", 64, 0, 64, 16, 32, 3, True, True),
    ("stats_narrow_16x16", 32, 0, 64, 16, 16, 3, False),
]


@pytest.mark.parametrize("case", STATS_CASES, ids=[c[0] for c in STATS_CASES])
def test_conv_epilogue_groupnorm_statistics(case):
    """stats_out of dsg_conv2d_fwd (per-tile sum / sum of squares of the tensor just written) gives the same
    GroupNorm scale/shift as a statistics pass over that tensor (dsg_gn_channel_stats), also across a concat."""
    name, c0, c1, cout, h, w, k, ups = case[:8]
    batch, cin, groups = 2, c0 + c1, 8
    x0 = _t(11, (batch, c0, h, w), 1.3).to(DEV)
    x1 = _t(12, (batch, c1, h, w)).to(DEV) if c1 else None
    wt = _t(13, (cout, cin, k, k), 1.0 / np.sqrt(cin * k * k)).to(DEV)
    bias = _t(14, (cout,), 0.5).to(DEV)   # non-zero channel means: the variance has something to cancel
    ho, wo = (2 * h, 2 * w) if ups else (h, w)
    res = _t(15, (batch, cout, ho, wo)).to(DEV)
    wr, wh = ops.relayout_conv_weight(wt), ops.relayout_conv_weight_h2(wt)
    whf = ops.relayout_conv_weight_h2_fold(wt) if len(case) > 8 else None
    y, st = ops.conv2d_fused(x0, wr, bias, src1=x1, ksize=k, upsample=ups, residual=res, cout=cout, weight_h2=wh,
                             weight_h2_fold=whf, want_stats=True)
    assert st is not None and st.shape[:2] == (batch, cout)
    tot = st.sum(dim=2).cpu()
    ref = torch.stack([y.double().sum(dim=(2, 3)), (y.double() ** 2).sum(dim=(2, 3))], dim=-1).cpu()
    assert torch.allclose(tot, ref, rtol=2e-6, atol=1e-4), float((tot - ref).abs().max())
    gamma, beta = (1 + _t(16, (cout,), 0.1)).to(DEV), _t(17, (cout,), 0.1).to(DEV)
    ss_ref = ops.gn_scale_shift(y, gamma, beta, groups, 1e-5)
    ss = ops.gn_scale_shift_from_parts(st, gamma, beta, groups, 1e-5, ho * wo)
    assert torch.allclose(ss, ss_ref, rtol=2e-6, atol=2e-6), float((ss - ss_ref).abs().max())
    # the same tensor as the second half of a concat whose first half has plain channel statistics
    z = _t(18, (batch, 24, ho, wo)).to(DEV)
    g2, b2 = (1 + _t(19, (24 + cout,), 0.1)).to(DEV), _t(20, (24 + cout,), 0.1).to(DEV)
    ss_ref2 = ops.gn_scale_shift(z, g2, b2, groups, 1e-5, src1=y)
    zst = torch.stack([z.double().sum(dim=(2, 3)), (z.double() ** 2).sum(dim=(2, 3))], dim=-1).unsqueeze(2).contiguous()
    ss2 = ops.gn_scale_shift_from_parts(zst, g2, b2, groups, 1e-5, ho * wo, stats1=st)
    assert torch.allclose(ss2, ss_ref2, rtol=2e-6, atol=2e-6), float((ss2 - ss_ref2).abs().max())
    # a conv the statistics kernel does not serve reports no tiles instead of writing nothing
    _, none = ops.conv2d_fused(x0, wr, bias, src1=x1, ksize=k, upsample=ups, cout=cout, want_stats=True)
    assert none is None


def _rand_conv_cases(n=24, seed=14555):
    """Seeded random draws over what the split-path dispatcher accepts: channel splits, padded cout tiles, tile-aligned
    and narrow maps, 3x3 / 1x1, folded / gathered up-sampling, residual / temb / norm combinations, batch 1..3."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        k = int(rng.choice([3, 3, 3, 1]))
        ups = bool(k == 3 and rng.random() < 0.25)
        c0 = int(rng.choice([16, 32, 48, 64, 96]))
        c1 = int(rng.choice([0, 0, 16, 32, 64]))
        cout = int(rng.choice([8, 24, 64, 72, 128, 192]))
        h, w = [(8, 8), (16, 16), (8, 32), (16, 32), (24, 32), (16, 64), (32, 32)][int(rng.integers(7))]
        if ups and w < 32 and 2 * w % 32:
            continue
        gn = bool(rng.random() < 0.6) and not ups
        out.append((c0, c1, cout, h, w, k, ups, gn, bool(rng.random() < 0.4) and k == 3, bool(rng.random() < 0.5),
                    int(rng.integers(1, 4)), bool(rng.random() < 0.5)))
    return out


@pytest.mark.parametrize("case", _rand_conv_cases(), ids=lambda c: "c%d+%d_o%d_%dx%d_k%d%s" % (c[:6] + ("u" if c[6] else "",)))
def test_conv_split_path_random_shapes(case):
    c0, c1, cout, h, w, k, ups, gn, temb, res, batch, fold = case
    cin = c0 + c1
    x0 = _t(41, (batch, c0, h, w), 1.5)
    x1 = _t(42, (batch, c1, h, w)) if c1 else None
    wt = _t(43, (cout, cin, k, k), 1.0 / np.sqrt(cin * k * k))
    bias = _t(44, (cout,), 0.1)
    gamma, beta = 1 + _t(45, (cin,), 0.1), _t(46, (cin,), 0.1)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    a = F.silu(F.group_norm(xin, 8, gamma, beta, 1e-5)) if gn else xin
    a = F.interpolate(a, scale_factor=2.0, mode="nearest") if ups else a
    want = F.conv2d(a.double(), wt.double(), bias.double(), padding=k // 2)
    mag = F.conv2d(a.double().abs(), wt.double().abs(), None, padding=k // 2) + 1e-30
    tproj = _t(47, (batch, cout + 3), 0.5)
    r = _t(48, tuple(want.shape))
    if temb:
        want = want + tproj[:, 1:1 + cout, None, None].double()
    if res:
        want = want + r.double()
    d = lambda t: None if t is None else t.to(DEV)
    wr, wh = ops.relayout_conv_weight(d(wt)), ops.relayout_conv_weight_h2(d(wt))
    whf = ops.relayout_conv_weight_h2_fold(d(wt)) if (ups and fold) else None
    ss = ops.gn_scale_shift(d(x0), d(gamma), d(beta), 8, 1e-5, src1=d(x1)) if gn else None
    tp = d(tproj)
    got, st = ops.conv2d_fused(d(x0), wr, d(bias), src1=d(x1), ksize=k, upsample=ups, gn_scale_shift=ss, silu=gn,
                               temb=tp[:, 1:] if temb else None, temb_stride=tp.stride(0), residual=d(r) if res else None,
                               cout=cout, weight_h2=wh, weight_h2_fold=whf, want_stats=True)
    err = ((got.cpu().double() - want).abs() / mag).max().item()
    assert err <= 6e-7, err   # fp32-class: a few ulps of sum |w||x| (plus the fast SiLU's staging error)
    if st is not None:  # whenever the kernel offers statistics they must be those of the tensor it wrote
        ref = torch.stack([got.double().sum(dim=(2, 3)), (got.double() ** 2).sum(dim=(2, 3))], dim=-1).cpu()
        assert torch.allclose(st.sum(dim=2).cpu(), ref, rtol=3e-6, atol=1e-4)


@pytest.mark.parametrize("cin,cout,res,gn", [(64, 64, True, True), (128, 64, False, True), (128, 128, True, True), (64, 128, False, False)])
def test_conv_two_workgroups_per_cu_is_bit_identical(cin, cout, res, gn):
    """The one-weight-slab kernel (8-row tiles, two workgroups per CU: dsg_set_tuning key 20) serves the cin <= 128 convs of
    grids with >= 512 workgroups.  Same K order, same summation tree of the statistics: its results and its per-tile
    GroupNorm partials are BITWISE those of the 16-row kernel."""
    from drivescenegen_amd import _lib
    lib = _lib.load()
    n, h, w = 4, 128, 128   # 4 x 16 x 4 x (cout / 64) 8-row tiles >= 512 for cout 128; cout 64: n = 8
    if cout == 64:
        n = 8
    d = lambda t: t.to(DEV)
    x = ops.to_blocked(d(_t(21, (n, cin, h, w))))
    wt = d(_t(22, (cout, cin, 3, 3), 0.05))
    bias = d(_t(23, (cout,)))
    r = ops.to_blocked(d(_t(24, (n, cout, h, w)))) if res else None
    ss = d(_t(25, (n, cin, 2))) if gn else None
    wr, wh = ops.relayout_conv_weight(wt), ops.relayout_conv_weight_h2(wt)
    outs = []
    try:
        for on in (0, 1):
            _lib.check(lib.dsg_set_tuning(20, on))
            y, st = ops.conv2d_fused(x, wr, bias, residual=r, gn_scale_shift=ss, silu=gn, weight_h2=wh, want_stats=True,
                                     src_blocked=True, dst_blocked=True)
            outs.append((y.clone(), st.clone()))
    finally:
        lib.dsg_set_tuning(20, 1)
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    ref = F.conv2d(F.silu(ops.from_blocked(x).cpu().double() * ss.cpu().double()[:, :, 0, None, None] + ss.cpu().double()[:, :, 1, None, None])
                   if gn else ops.from_blocked(x).cpu().double(), wt.cpu().double(), bias.cpu().double(), padding=1)
    if res:
        ref = ref + ops.from_blocked(r).cpu().double()
    assert rel_l2(ops.from_blocked(outs[1][0]).cpu(), ref) < 2e-6
