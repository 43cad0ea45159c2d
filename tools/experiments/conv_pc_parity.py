"""The wave-specialised persistent form of the shallow-level 3x3 conv (csrc/conv_pc.hip; dsg_set_tuning key 28).

One workgroup per CU walks tiles for the whole launch: producer waves fetch, normalise, activate and split the halo
patches three K-chunks ahead, consumer waves issue the MFMAs and run the epilogue (the resnet convs of the network
train.py:39-57 builds at 64 / 128 channels, evaluated at training_pipeline.py:84 and inside DDPMPipeline.__call__).
It must give the bits of the kernels it replaces -- same operands, same accumulation order, same epilogue and the same
GroupNorm statistics -- for single and concatenated sources, with and without residual / time embedding, for cout tiles
that are not full, across image boundaries inside one workgroup's walk; and the fp32-class accuracy of the split path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import _lib, ops, synth  # noqa: E402
from tests.common import CFG2, noisy_inputs, rel_l2, synth_weights  # noqa: E402

DEV = "cuda"


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))


class pc_kernel:
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        _lib.check(_lib.load().dsg_set_tuning(28, int(self.on)))

    def __exit__(self, *exc):
        _lib.load().dsg_set_tuning(28, 0)   # (the library's default: measured slower than the kernels it replaces)


# (n, c0, c1, cout, h, w, residual, temb)
SHAPES = {
    "64_256": (16, 64, 0, 64, 256, 256, True, False),
    "64_256_temb": (5, 64, 0, 64, 256, 256, False, True),        # odd image count: tile lists that do not divide by 8
    "128_128": (16, 128, 0, 128, 128, 128, True, False),
    "cat_96_to_64": (8, 64, 32, 64, 256, 256, False, True),
    "128_to_96_tail": (16, 128, 0, 96, 128, 128, False, False),    # second cout tile half empty
    "32_rows_wide": (16, 64, 0, 64, 64, 512, True, True),
}


@pytest.mark.parametrize("name", list(SHAPES))
def test_conv_pc_is_bitwise_the_kernels_it_replaces(name):
    n, c0, c1, cout, h, w, has_res, has_temb = SHAPES[name]
    c = c0 + c1
    g = lambda t: None if t is None else t.to(DEV)
    x0, x1 = _t(1, (n, c0, h, w)), (_t(2, (n, c1, h, w)) if c1 else None)
    for i in range(n):   # every image its own scale / shift, so that a stale table shows
        x0[i] *= 1.0 + 0.1 * i
    wt, b = _t(3, (cout, c, 3, 3), 1.0 / np.sqrt(9 * c)), _t(4, (cout,), 0.1)
    ss = torch.stack([1 + _t(5, (n, c), 0.2), _t(6, (n, c), 0.3)], -1).contiguous()
    temb = _t(7, (n, cout), 0.3) if has_temb else None
    res = _t(8, (n, cout, h, w)) if has_res else None
    x0b = ops.to_blocked(g(x0))
    x1b = ops.to_blocked(g(x1)) if x1 is not None else None
    resb = ops.to_blocked(g(res)) if res is not None else None
    kw = dict(src1=x1b, ksize=3, cout=cout, gn_scale_shift=g(ss), silu=True, temb=g(temb), temb_stride=cout, src_blocked=True,
              dst_blocked=True, weight_h2=ops.relayout_conv_weight_h2(g(wt)), residual=resb, want_stats=True)
    with pc_kernel(True):
        ya, sa = ops.conv2d_fused(x0b, None, g(b), **kw)
    with pc_kernel(False):
        yb, sb = ops.conv2d_fused(x0b, None, g(b), **kw)
    assert torch.isfinite(ya).all()
    assert torch.equal(ya, yb), float((ya - yb).abs().max())
    assert sa is not None and sb is not None and torch.equal(sa, sb)
    if name in ("64_256", "cat_96_to_64"):   # and both are the fp32-class evaluation of the reference expression
        x = (x0 if x1 is None else torch.cat([x0, x1], 1))[:2].double()
        act = F.silu(x * ss[:2].double()[:, :, 0, None, None] + ss[:2].double()[:, :, 1, None, None])
        ref = F.conv2d(act, wt.double(), b.double(), padding=1)
        if temb is not None:
            ref = ref + temb[:2].double()[:, :, None, None]
        if res is not None:
            ref = ref + res[:2].double()
        scale = F.conv2d(act.abs(), wt.double().abs(), padding=1) + 1.0
        got = ops.from_blocked(ya[:2]).cpu().double()
        assert ((got - ref).abs() / scale).max().item() <= 6e-7


def test_whole_network_with_and_without_conv_pc_is_bitwise():
    net = synth_weights(d.UNet2DModel(**CFG2)).to(DEV).eval().requires_grad_(False)
    x = noisy_inputs(CFG2, 16).to(DEV)
    t = torch.full((16,), 500, dtype=torch.int64, device=DEV)
    with pc_kernel(True):
        y1 = net(x, t).sample.clone()
    with pc_kernel(False):
        y0 = net(x, t).sample.clone()
    assert torch.isfinite(y1).all()
    assert torch.equal(y0, y1), float((y0 - y1).abs().max())
