"""Calibration: what this box's HBM delivers to plain streaming kernels on a bench-sized activation tensor
(16 x 64 x 256 x 256 fp32 = 268 MB): copy, a + b, and the library's own GroupNorm-apply pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops

x = torch.randn(16, 64, 256, 256, device="cuda")
y = torch.randn_like(x)
o = torch.empty_like(x)
ss = torch.randn(16, 64, 2, device="cuda")


def timeit(name, f, nbytes, iters=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:28s} {ms*1e3:8.1f} us  {nbytes/ms/1e9:6.2f} TB/s")


n = x.numel() * 4
timeit("copy (1R + 1W)", lambda: o.copy_(x), 2 * n)
timeit("add  (2R + 1W)", lambda: torch.add(x, y, out=o), 3 * n)
timeit("sum  (1R)", lambda: x.sum(), n)
timeit("fill (1W)", lambda: o.fill_(1.0), n)
if hasattr(ops, "gn_apply"):
    timeit("dsg gn_apply (1R + 1W)", lambda: ops.gn_apply(x, ss, True, out=o) if "out" in ops.gn_apply.__code__.co_varnames else ops.gn_apply(x, ss, True), 2 * n)
