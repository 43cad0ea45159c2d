"""conv_in / conv_out shapes: matrix-core kernel vs the direct VALU kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops
B = 16
def run(name, cin, cout, gn, direct):
    x = torch.randn(B, cin, 256, 256, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wr = ops.relayout_conv_weight(w); bias = torch.randn(cout, device="cuda")
    ss = torch.randn(B, cin, 2, device="cuda") if gn else None
    out = torch.empty(B, cout, 256, 256, device="cuda")
    f = lambda: ops.conv2d_fused(x, wr, bias, gn_scale_shift=ss, silu=gn, out=out, direct=direct, cout=cout)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name} direct={direct}: {e0.elapsed_time(e1)/20:.3f} ms")
for d in (False, True):
    run("conv_in 4->64", 4, 64, False, d)
    run("conv_out 64->4", 64, 4, True, d)
