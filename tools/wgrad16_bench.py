"""Time of the mixed-precision 3x3 weight gradient (conv_wgrad16_kernel + its reduce pass) on the network's layer shapes.
Usage: wgrad16_bench.py [bf16|fp16] [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops
dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for (cin, cout, h, w) in [(64, 64, 256, 256), (128, 128, 128, 128), (256, 256, 64, 64), (512, 512, 32, 32), (1024, 512, 32, 32)]:
    x = torch.randn(b, cin // 8, h, w, 8, device="cuda").to(dt)
    dy = (torch.randn(b, cout // 8, h, w, 8, device="cuda") * 1e-2).to(dt)
    ss = torch.stack([1 + 0.1 * torch.randn(b, cin, device="cuda"), 0.1 * torch.randn(b, cin, device="cuda")], -1).contiguous()
    dw = torch.zeros(cout, cin, 3, 3, device="cuda")
    for _ in range(3): ops.conv_wgrad(x, dy, dw, ksize=3, gn_scale_shift=ss, silu=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.conv_wgrad(x, dy, dw, ksize=3, gn_scale_shift=ss, silu=True)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"  {cin}->{cout} @{h}x{w} b{b}: {t*1e6:.0f} us  {2*b*h*w*cin*cout*9/t/1e12:.0f} TF/s")
