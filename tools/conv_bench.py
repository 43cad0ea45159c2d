"""Micro-benchmark of dsg_conv2d_fwd on the layer shapes of the default U-Net (B=16): TF/s per shape.
Usage: python tools/conv_bench.py [iters]"""
import os, sys
os.environ.setdefault("DSG_TESTING", "1")  # dsg_set_tuning is a test hook
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops

B = int(os.environ.get("B", 16))
from drivescenegen_amd import _lib
if os.environ.get("DSG_VARIANT"):
    _lib.check(_lib.load().dsg_set_tuning(0, int(os.environ["DSG_VARIANT"])))
    print("conv variant", os.environ["DSG_VARIANT"])
if os.environ.get("DSG_ROWS"):
    _lib.check(_lib.load().dsg_set_tuning(3, int(os.environ["DSG_ROWS"])))
    print("h2 rows/wave", os.environ["DSG_ROWS"])
if os.environ.get("DSG_KC"):
    _lib.check(_lib.load().dsg_set_tuning(1, int(os.environ["DSG_KC"])))
    print("kc", os.environ["DSG_KC"])
if os.environ.get("DSG_WAVES"):
    _lib.check(_lib.load().dsg_set_tuning(6, int(os.environ["DSG_WAVES"])))
    print("h2 waves", os.environ["DSG_WAVES"])
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
# name, c0, c1, cout, h, k, stride, ups, gn, res
SHAPES = [
    ("res64@256 conv1 (gn,temb)", 64, 0, 64, 256, 3, 1, False, True, False),
    ("res64@256 conv2 (gn,res)", 64, 0, 64, 256, 3, 1, False, True, True),
    ("res128@128", 128, 0, 128, 128, 3, 1, False, True, True),
    ("res256@64", 256, 0, 256, 64, 3, 1, False, True, True),
    ("res512@32", 512, 0, 512, 32, 3, 1, False, True, True),
    ("up 1024->512@32 cat", 512, 512, 512, 32, 3, 1, False, True, False),
    ("up 192->64@256 cat", 128, 64, 64, 256, 3, 1, False, True, False),
    ("upsample 512@32->64", 512, 0, 512, 32, 3, 1, True, False, False),
    ("upsample 128@128->256", 128, 0, 128, 128, 3, 1, True, False, False),
    ("down s2 64@256", 64, 0, 64, 256, 3, 2, False, False, False),
    ("1x1 192->64@256", 128, 64, 64, 256, 1, 1, False, False, False),
    ("1x1 1024->512@32", 512, 512, 512, 32, 1, 1, False, False, False),
    ("qkv 512->1536@32", 512, 0, 1536, 32, 1, 1, False, True, False),
]
tot_t = tot_f = 0.0
if os.environ.get('ONLY'):
    SHAPES = [s for s in SHAPES if os.environ['ONLY'] in s[0]]
for name, c0, c1, cout, h, k, s, ups, gn, res in SHAPES:
    cin = c0 + c1
    if os.environ.get("NOGN") == "1":   # ablation: the same conv without the GroupNorm-apply + SiLU staging work
        gn = False
    if os.environ.get("NORES") == "1":
        res = False
    x0 = torch.randn(B, c0, h, h, device=dev)
    x1 = torch.randn(B, c1, h, h, device=dev) if c1 else None
    w = torch.randn(cin, k * k, cout, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    ss = torch.randn(B, cin, 2, device=dev) if gn else None
    ho = (2 * h if ups else h) // s
    r = torch.randn(B, cout, ho, ho, device=dev) if res else None
    out = torch.empty(B, cout, ho, ho, device=dev)
    wh = None
    if os.environ.get("H2") == "1" and s == 1 and cout % 64 == 0:
        wh = ops.relayout_conv_weight_h2(torch.randn(cout, cin, k, k, device=dev) * 0.05)
    blk = os.environ.get("BLOCKED") == "1" and not (ups and k == 3 and True) and s == 1 and cout % 8 == 0
    if blk:  # channel-blocked tensors (same bytes, [N, C/8, H, W, 8])
        x0, x1, r, out = ops.to_blocked(x0), (ops.to_blocked(x1) if c1 else None), (ops.to_blocked(r) if res else None), ops.to_blocked(out)
    f = lambda: ops.conv2d_fused(x0, w, bias, src1=x1, ksize=k, stride=s, upsample=ups, gn_scale_shift=ss, silu=gn,
                                 residual=r, out=out, weight_h2=wh, src_blocked=blk, dst_blocked=blk)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * ho * ho * cout * cin * k * k
    tot_t += ms
    tot_f += fl
    print(f"{name:28s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s")
print(f"{'TOTAL':28s} {tot_t:8.3f} ms  {tot_f / tot_t / 1e9:7.1f} TF/s")
