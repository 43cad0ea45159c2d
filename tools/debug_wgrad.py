import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from drivescenegen_amd import ops
torch.manual_seed(0)
B, cin, cout, h, w = int(os.environ.get("B", 1)), 32, 64, int(os.environ.get("H", 4)), 32
x = torch.randn(B, cin, h, w); dy = torch.randn(B, cout, h, w)
wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
y = F.conv2d(x, wt, None, padding=1); y.backward(dy)
dw = torch.zeros(cout, cin, 3, 3, device="cuda")
ops.conv_wgrad(x.cuda(), dy.cuda(), dw, ksize=3)
got = dw.cpu(); want = wt.grad
for tp in range(9):
    g, r = got[:, :, tp // 3, tp % 3], want[:, :, tp // 3, tp % 3]
    print(tp, "rel err", float((g - r).norm() / r.norm()), "ratio", float((g * r).sum() / (r * r).sum()))
