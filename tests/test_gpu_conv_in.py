"""conv_in (csrc/conv_in.hip): fp32 [N, C<=8, H, W] image -> channel-blocked activations in every arithmetic mode, with the
per-tile GroupNorm statistics of the result.  Reference layer: UNet2DModel.conv_in = nn.Conv2d(in_channels, boc[0], 3,
padding=1) (train.py:39-57 sets in_channels); the comparison is torch's conv2d in fp64 on the CPU.
Tolerances: fp32-equivalent mode: |err| <= 6e-7 * sum|w||x| per output (the fmaf chain's own bound is ~72 * 6e-8 of it);
bf16 / fp16: exact conv of the once-rounded operands, then one output rounding (2^-8 / 2^-11 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from drivescenegen_amd import _lib, ops, synth  # noqa: E402

DEV = "cuda"


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


def _run(x, w, b, dtype=0, stats=False):
    wr = ops.relayout_conv_weight(w.to(DEV))
    y, st = ops.conv2d_fused(x.to(DEV), wr, b.to(DEV) if b is not None else None, ksize=3, cout=w.shape[0],
                             dst_blocked=True, compute_dtype=dtype, want_stats=True)
    torch.cuda.synchronize()
    return ops.from_blocked(y).float().cpu(), (st.cpu() if st is not None else None)


CASES = [(3, 32, 16, 32, 2), (4, 64, 32, 64, 2), (8, 64, 16, 96, 1), (4, 128, 48, 32, 3), (1, 96, 16, 32, 1)]


@pytest.mark.parametrize("cin,cout,h,w,n", CASES, ids=lambda v: str(v))
def test_conv_in_fp32_equivalent_vs_fp64(cin, cout, h, w, n):
    x, wt, b = _t(1, (n, cin, h, w), 1.7), _t(2, (cout, cin, 3, 3), 0.3), _t(3, (cout,), 0.5)
    got, st = _run(x, wt, b, stats=True)
    want = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    bound = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1) + b.double().abs().view(1, -1, 1, 1)
    assert torch.isfinite(got).all()
    assert ((got.double() - want).abs() <= 6e-7 * bound + 1e-30).all(), float(((got.double() - want).abs() / bound).max())
    # the statistics are those of the tensor written, tile by tile (16 x 32 pixels)
    assert st is not None and st.shape == (n, cout, (h // 16) * (w // 32), 2)
    ref = torch.stack([got.double().sum(dim=(2, 3)), (got.double() ** 2).sum(dim=(2, 3))], dim=-1)
    assert torch.allclose(st.sum(dim=2), ref, rtol=3e-6, atol=1e-4), float((st.sum(dim=2) - ref).abs().max())
    t0 = got[:, :, :16, :32].double()
    assert torch.allclose(st[:, :, 0, 0], t0.sum(dim=(2, 3)), rtol=3e-6, atol=1e-4)
    # same values as the exact f32-MFMA kernel that served the call before (tuning key 21 switches this kernel off)
    lib = _lib.load()
    _lib.check(lib.dsg_set_tuning(21, 0))
    try:
        old, st_old = _run(x, wt, b)
    finally:
        _lib.check(lib.dsg_set_tuning(21, 1))
    assert st_old is None  # (that kernel writes no statistics: the plan then runs a pass of its own)
    assert ((got - old).abs().double() <= 6e-7 * bound).all()
    assert not torch.equal(got, old)


def test_conv_in_needs_no_range_guard():
    """Operand scaling is built in: a 1e5-scale image, output channels whose weights differ by 1e9, tiny images."""
    cin, cout, h, w = 4, 64, 16, 64
    x = _t(4, (2, cin, h, w), 1.0)
    x[0] *= 1e5
    x[1] *= 3e-7
    wt = _t(5, (cout, cin, 3, 3), 0.3)
    wt[::3] *= 1e-6
    wt[1::3] *= 2e3
    got, _ = _run(x, wt, None)
    want = F.conv2d(x.double(), wt.double(), None, padding=1)
    bound = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1)
    assert torch.isfinite(got).all()
    assert ((got.double() - want).abs() <= 4e-7 * bound).all(), float(((got.double() - want).abs() / bound).max())
    # zeros stay zeros; an all-zero patch does not divide by its maximum
    z, _ = _run(torch.zeros(1, cin, h, w), wt, None)
    assert (z == 0).all()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout", [(8, 64), (3, 32), (4, 128)])
def test_conv_in_16bit_modes(dtype, cin, cout):
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    n, h, w = 2, 32, 64
    x, wt, b = _t(6, (n, cin, h, w), 1.5), _t(7, (cout, cin, 3, 3), 0.3), _t(8, (cout,), 0.5)
    got, st = _run(x, wt, b, dtype=dtype, stats=True)
    want = F.conv2d(x.to(td).double(), wt.to(td).double(), b.double(), padding=1)
    bound = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1) + b.double().abs().view(1, -1, 1, 1)
    eps = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11
    err = (got.double() - want).abs()
    assert (err <= eps * want.abs() + 3e-7 * bound).all(), float((err / bound).max())
    ref = torch.stack([got.double().sum(dim=(2, 3)), (got.double() ** 2).sum(dim=(2, 3))], dim=-1)
    assert st is not None and torch.allclose(st.sum(dim=2), ref, rtol=3e-6, atol=1e-4)


def test_shapes_outside_the_tiling_keep_their_old_kernels():
    x, wt = _t(9, (1, 4, 8, 32), 1.0), _t(10, (64, 4, 3, 3), 0.3)  # 8 rows: not a 16 x 32 tile
    got, st = _run(x, wt, None)
    want = F.conv2d(x.double(), wt.double(), None, padding=1)
    assert st is None and (got.double() - want).abs().max() <= 1e-5
