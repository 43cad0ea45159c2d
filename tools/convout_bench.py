import os, sys, torch
sys.path.insert(0, "/root/repo")
from drivescenegen_amd import ops
B=16
x = torch.randn(B, 64, 256, 256, device="cuda"); w = torch.randn(4, 64, 3, 3, device="cuda") * 0.05
wr = ops.relayout_conv_weight(w); bias = torch.randn(4, device="cuda"); ss = torch.randn(B, 64, 2, device="cuda")
out = torch.empty(B, 4, 256, 256, device="cuda")
f = lambda: ops.conv2d_fused(x, wr, bias, gn_scale_shift=ss, silu=True, out=out, cout=4)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize(); print("conv_out 64->4 B=16:", e0.elapsed_time(e1)/20, "ms")
