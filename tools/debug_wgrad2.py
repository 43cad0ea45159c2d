import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from drivescenegen_amd import ops
torch.manual_seed(0)
B, cin, cout, h, w = 1, 32, 64, 4, 32
x0 = torch.randn(B, cin, h, w); dy = torch.randn(B, cout, h, w)
for row in range(h):
    for colset in ("all", "c0", "c31", "mid"):
        x = torch.zeros_like(x0)
        if colset == "all": x[:, :, row, :] = x0[:, :, row, :]
        elif colset == "c0": x[:, :, row, 0] = x0[:, :, row, 0]
        elif colset == "c31": x[:, :, row, 31] = x0[:, :, row, 31]
        else: x[:, :, row, 8:16] = x0[:, :, row, 8:16]
        wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
        F.conv2d(x, wt, None, padding=1).backward(dy)
        dw = torch.zeros(cout, cin, 3, 3, device="cuda")
        ops.conv_wgrad(x.cuda(), dy.cuda(), dw, ksize=3)
        e = [(float((dw.cpu()[:, :, t // 3, t % 3] - wt.grad[:, :, t // 3, t % 3]).norm() / (wt.grad[:, :, t // 3, t % 3].norm() + 1e-30))) for t in range(9)]
        print(row, colset, " ".join(f"{v:.0e}" for v in e))
