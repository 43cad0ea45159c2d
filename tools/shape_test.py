import os, sys, torch
sys.path.insert(0, "/root/repo")
from drivescenegen_amd import ops
def run(B, c, h, iters=20):
    x = torch.randn(B, c, h, h, device="cuda"); w = torch.randn(c, c, 3, 3, device="cuda") * 0.05
    wr, wh = ops.relayout_conv_weight(w), ops.relayout_conv_weight_h2(w)
    ss = torch.randn(B, c, 2, device="cuda"); r = torch.randn(B, c, h, h, device="cuda"); out = torch.empty_like(r)
    f = lambda: ops.conv2d_fused(x, wr, None, gn_scale_shift=ss, silu=True, residual=r, out=out, weight_h2=wh)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"B={B} c={c} h={h}: {ms:.3f} ms  {2.0*B*h*h*c*c*9/ms/1e9:.1f} TF/s  {(3*B*c*h*h*4)/ms/1e6:.0f} GB/s")
run(16, 64, 256); run(64, 64, 128); run(256, 64, 64); run(1024, 64, 32); run(4, 64, 512)
