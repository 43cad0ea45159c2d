"""One layer shape of the 16-bit 3x3 weight gradient, two calls with 64 x 64 workgroups and two with the wide ones (for the
counter passes of tools/pmc_wgrad16.sh).  Usage: DSG_TESTING=1 wgrad16_one.py cin cout h w [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops, _lib
cin, cout, h, w = (int(v) for v in sys.argv[1:5])
b = int(sys.argv[5]) if len(sys.argv) > 5 else 32
lib = _lib.load()
x = torch.randn(b, cin // 8, h, w, 8, device="cuda").to(torch.bfloat16)
dy = (torch.randn(b, cout // 8, h, w, 8, device="cuda") * 1e-2).to(torch.bfloat16)
ss = torch.stack([1 + 0.1 * torch.randn(b, cin, device="cuda"), 0.1 * torch.randn(b, cin, device="cuda")], -1).contiguous()
dw = torch.zeros(cout, cin, 3, 3, device="cuda")
for wide in (0, 0, 1, 1):
    _lib.check(lib.dsg_set_tuning(29, wide))
    ops.conv_wgrad(x, dy, dw, ksize=3, gn_scale_shift=ss, silu=True)
torch.cuda.synchronize()
