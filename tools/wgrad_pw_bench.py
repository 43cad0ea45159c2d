"""Time of the fp32 pointwise (1x1) weight gradient (fp16x2-split kernel + reduce pass) on a few layer shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops
for (c0, c1, cout, h, w, b) in [(512, 512, 512, 32, 32, 64), (512, 512, 512, 34, 32, 64), (512, 512, 512, 64, 32, 32), (512,512,512,66,32,32), (128, 64, 64, 256, 256, 16)]:
    x0 = torch.randn(b, c0, h, w, device="cuda"); x1 = torch.randn(b, c1, h, w, device="cuda") if c1 else None
    dy = torch.randn(b, cout, h, w, device="cuda")
    dw = torch.zeros(cout, c0 + c1, 1, 1, device="cuda")
    for _ in range(3): ops.conv_wgrad(x0, dy, dw, src1=x1, ksize=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.conv_wgrad(x0, dy, dw, src1=x1, ksize=1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"  {c0}+{c1}->{cout} @{h}x{w} b{b}: {dt*1e6:.0f} us  {2*b*h*w*(c0+c1)*cout/dt/1e12:.0f} TF/s-eq")
