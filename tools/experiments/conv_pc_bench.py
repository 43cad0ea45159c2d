"""Time of one shallow-level fp32-equivalent 3x3 conv with the producer / consumer kernel on (key 28 = 1) and off.
Usage: conv_pc_bench.py [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DSG_TESTING", "1")
import torch
from drivescenegen_amd import ops, _lib
b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = _lib.load()
for (cin, cout, h, w, res) in [(64, 64, 256, 256, True), (128, 128, 128, 128, True), (64, 128, 128, 128, False)]:
    x = ops.to_blocked(torch.randn(b, cin, h, w, device="cuda"))
    wt = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    wh = ops.relayout_conv_weight_h2(wt)
    ss = torch.stack([1 + 0.1 * torch.randn(b, cin, device="cuda"), 0.1 * torch.randn(b, cin, device="cuda")], -1).contiguous()
    r = ops.to_blocked(torch.randn(b, cout, h, w, device="cuda")) if res else None
    bias = torch.zeros(cout, device="cuda")
    for on in (0, 1):
        _lib.check(lib.dsg_set_tuning(28, on))
        f = lambda: ops.conv2d_fused(x, None, bias, ksize=3, cout=cout, gn_scale_shift=ss, silu=True, src_blocked=True, dst_blocked=True,
                                     weight_h2=wh, residual=r, want_stats=True)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
        print(f"  {cin}->{cout} @{h}x{w} b{b} pc={on}: {t*1e6:.0f} us  {2*b*h*w*cin*cout*9/t/1e12:.0f} TF/s-eq")
