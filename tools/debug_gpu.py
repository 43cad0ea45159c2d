import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import drivescenegen_amd as d
from drivescenegen_amd import ops, synth
from oracle.scheduler_oracle import OracleDDPMScheduler
from oracle.unet_oracle import timestep_embedding
DEV = "cuda"
def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))
o, s = OracleDDPMScheduler(), d.DDPMScheduler()
x0, nz = _t(51, (4, 3, 64, 64)).clamp(-1, 1), _t(52, (4, 3, 64, 64))
t = torch.tensor([0, 3, 500, 999])
got = s.add_noise(x0.to(DEV), nz.to(DEV), t.to(DEV)).cpu()
want = o.add_noise(x0, nz, t)
diff = (got - want).abs()
print("add_noise maxdiff", diff.max().item(), "n mismatched", (diff > 0).sum().item(), "of", diff.numel())
for n in range(4):
    print("  sample", n, "mismatch", (diff[n] > 0).sum().item())
sa, sb = s._sqrt_tables(torch.device(DEV))
print("tables on device equal:", torch.equal(sa.cpu(), o.alphas_cumprod ** 0.5), torch.equal(sb.cpu(), (1 - o.alphas_cumprod) ** 0.5))
# time embedding
ch, dim = 64, 256
tt = torch.tensor([0, 1, 499, 999, 37], dtype=torch.long)
w1, b1 = _t(41, (dim, ch), 1 / 8.0), _t(42, (dim,), 0.1)
w2, b2 = _t(43, (dim, dim), 1 / 16.0), _t(44, (dim,), 0.1)
emb = timestep_embedding(tt, ch)
ref = F.silu(F.linear(F.silu(F.linear(emb, w1, b1)), w2, b2))
got = ops.time_embed(tt.to(DEV), w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV)).cpu()
print("temb max abs", (got - ref).abs().max().item(), "rel", ((got - ref).norm() / ref.norm()).item(), "per-row", (got - ref).abs().amax(1))
ref64 = F.silu(F.linear(F.silu(F.linear(timestep_embedding(tt, ch).double(), w1.double(), b1.double())), w2.double(), b2.double()))
print("  cpu32 vs 64", (ref - ref64).abs().max().item(), " gpu vs 64", (got - ref64).abs().max().item())
# conv fp64
x = _t(11, (2, 256, 32, 32)); wt = _t(12, (64, 256, 3, 3), 1 / 48.0)
ref64 = F.conv2d(x.double(), wt.double(), None, padding=1)
cpu32 = F.conv2d(x, wt, None, padding=1)
g = ops.conv2d_fused(x.to(DEV), ops.relayout_conv_weight(wt.to(DEV))).cpu()
print("conv: gpu-vs-64", (g - ref64).abs().max().item(), "cpu32-vs-64", (cpu32 - ref64).abs().max().item(), "rms gpu", (g - ref64).pow(2).mean().sqrt().item(), "rms cpu", (cpu32 - ref64).pow(2).mean().sqrt().item())
