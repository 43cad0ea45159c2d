"""GroupNorm backward on channel-blocked 16-bit tensors (dsg_gn_bwd_blocked) at the U-Net's levels: time and HBM rate
against the 5 tensor passes it makes (x and dy read twice, dx written).  Usage: gn_bwd_bench.py [batch] [bf16|fp16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
tdt = torch.bfloat16 if dt == "bf16" else torch.float16
for c, h in ((64, 256), (128, 128), (256, 64), (512, 32), (1024, 32)):
    x = torch.randn(B, c // 8, h, h, 8, device="cuda").to(tdt)
    dy = torch.randn_like(x)
    ss = torch.randn(B, c, 2, device="cuda")
    mr = torch.rand(B, c, 2, device="cuda") + 0.5
    gamma = torch.randn(c, device="cuda")
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    f = lambda: ops.gn_bwd_blocked(x, dy, ss, mr, gamma, 32, True, dg, db)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = 5 * x.numel() * 2
    print(f"C={c:5d} {h}x{h} B={B}: {ms * 1e3:8.1f} us  {by / ms / 1e9:6.2f} TB/s over 5 passes of {x.numel() * 2 / 1e6:.0f} MB")
