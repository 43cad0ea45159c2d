"""conv_out (csrc/conv_out.hip): channel-blocked activations -> GroupNorm scale/shift + SiLU -> 3x3 conv -> fp32
[N, C<=8, H, W] image, in every arithmetic mode.  Reference layers: UNet2DModel.conv_norm_out, conv_act (SiLU), conv_out
(train.py:39-57 sets out_channels); the comparison is torch in fp64 on the CPU.
Tolerances: fp32-equivalent: |err| <= 1e-6 * sum|w||a| (a = the activated input; the hardware exp / reciprocal of SiLU are
good to ~2 ulp); bf16 / fp16: the conv of the once-rounded activated input and weights (2^-8 / 2^-11 relative per operand)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from drivescenegen_amd import _lib, ops, synth  # noqa: E402

DEV = "cuda"


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


def _case(cin, cout, h, w, n, dtype=0, wscale=None):
    x = _t(1, (n, cin, h, w), 1.4) + 0.3
    gamma, beta = 1 + _t(2, (cin,), 0.2), _t(3, (cin,), 0.2)
    wt, b = _t(4, (cout, cin, 3, 3), 0.1), _t(5, (cout,), 0.3)
    if wscale is not None:
        wt = wt * torch.tensor(wscale, dtype=torch.float32).view(-1, 1, 1, 1)
    td = {0: torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    xq = x.to(td)  # what the blocked tensor holds
    ss = ops.gn_scale_shift(xq.float().to(DEV), gamma.to(DEV), beta.to(DEV), 8, 1e-5)
    wr = ops.relayout_conv_weight(wt.to(DEV))
    xb = ops.to_blocked(x.to(DEV), dtype)
    got = ops.conv2d_fused(xb, wr, b.to(DEV), ksize=3, cout=cout, gn_scale_shift=ss, silu=True, src_blocked=True,
                           compute_dtype=dtype)
    torch.cuda.synchronize()
    ssd = ss.double().cpu()
    a = F.silu(xq.double() * ssd[..., 0].view(n, cin, 1, 1) + ssd[..., 1].view(n, cin, 1, 1))
    return got.cpu(), a, wt, b


CASES = [(64, 4, 16, 32, 2), (64, 3, 32, 64, 1), (32, 3, 64, 64, 2), (64, 8, 16, 64, 2), (16, 1, 16, 32, 1)]


@pytest.mark.parametrize("cin,cout,h,w,n", CASES, ids=lambda v: str(v))
def test_conv_out_fp32_equivalent_vs_fp64(cin, cout, h, w, n):
    got, a, wt, b = _case(cin, cout, h, w, n)
    want = F.conv2d(a, wt.double(), b.double(), padding=1)
    bound = F.conv2d(a.abs(), wt.double().abs(), None, padding=1) + b.double().abs().view(1, -1, 1, 1)
    assert got.shape == want.shape and torch.isfinite(got).all()
    err = (got.double() - want).abs()
    assert (err <= 1e-6 * bound).all(), float((err / bound).max())
    # the kernel that served the call before (tuning key 22 switches this one off) gives the same values to round-off
    lib = _lib.load()
    _lib.check(lib.dsg_set_tuning(22, 0))
    try:
        old, *_ = _case(cin, cout, h, w, n)
    finally:
        _lib.check(lib.dsg_set_tuning(22, 1))
    assert ((got - old).abs().double() <= 2e-6 * bound).all()
    assert not torch.equal(got, old)


def test_conv_out_weight_scales_per_output_channel():
    got, a, wt, b = _case(64, 4, 16, 32, 2, wscale=[1e-6, 1.0, 3e3, 1e-3])
    want = F.conv2d(a, wt.double(), b.double(), padding=1)
    bound = F.conv2d(a.abs(), wt.double().abs(), None, padding=1) + b.double().abs().view(1, -1, 1, 1)
    err = (got.double() - want).abs()
    assert torch.isfinite(got).all() and (err <= 1e-6 * bound).all(), float((err / bound).max())


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout", [(64, 8), (64, 4), (32, 3)])
def test_conv_out_16bit_modes(dtype, cin, cout):
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    got, a, wt, b = _case(cin, cout, 32, 64, 2, dtype=dtype)
    want = F.conv2d(a.to(td).double(), wt.to(td).double(), b.double(), padding=1)
    bound = F.conv2d(a.abs(), wt.double().abs(), None, padding=1) + b.double().abs().view(1, -1, 1, 1)
    eps = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11
    err = (got.double() - want).abs()
    # (an activated value that sits on a rounding boundary may round the other way after the hardware exp: one operand ulp)
    assert (err <= 0.25 * eps * bound + 1e-6 * bound).all(), float((err / bound).max())
    assert float(err.mean() / bound.mean()) <= 0.02 * eps
