"""Micro-benchmark of dsg_attention_fwd on the bench net's mid-block shape (B=16, 64 heads x 8, 1024 tokens) and
cfg4's (B=8, 32 heads x 8 at 256 tokens, 64 x 8 at 1024): DSG_TUNING=14=0 selects the VALU kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops

for n, c, heads, l in ((16, 512, 64, 1024), (8, 256, 32, 1024), (8, 512, 64, 256), (1, 512, 64, 1024)):
    qkv = torch.randn(n, 3 * c, l, device="cuda")
    for _ in range(3):
        ops.attention(qkv, heads)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention(qkv, heads)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    pairs = n * heads * l * l
    print(f"n={n} heads={heads} l={l}: {ms*1e3:8.1f} us  {pairs/ms/1e6:7.1f} G pairs/s")
