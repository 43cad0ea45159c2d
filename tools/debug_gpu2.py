import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import drivescenegen_amd as d
from drivescenegen_amd import synth
from oracle.scheduler_oracle import OracleDDPMScheduler
DEV = "cuda"
def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32))
o, s = OracleDDPMScheduler(), d.DDPMScheduler()
x0, nz = _t(51, (4, 3, 64, 64)).clamp(-1, 1), _t(52, (4, 3, 64, 64))
t = torch.tensor([0, 3, 500, 999])
got = s.add_noise(x0.to(DEV), nz.to(DEV), t.to(DEV)).cpu().numpy()
want = o.add_noise(x0, nz, t).numpy()
ac = o.alphas_cumprod
sa = (ac**0.5)[t].numpy(); sb = ((1-ac)**0.5)[t].numpy()
m1 = (sa[:,None,None,None]*x0.numpy()).astype(np.float32); m2=(sb[:,None,None,None]*nz.numpy()).astype(np.float32)
sep = (m1+m2).astype(np.float32)
x64 = x0.numpy().astype(np.float64); n64 = nz.numpy().astype(np.float64)
f1 = (sa[:,None,None,None].astype(np.float64)*x64 + m2.astype(np.float64)).astype(np.float32)
f2 = (m1.astype(np.float64) + sb[:,None,None,None].astype(np.float64)*n64).astype(np.float32)
print("cpu oracle == separate:", np.array_equal(want, sep))
print("gpu == separate:", np.array_equal(got, sep), " gpu == fma(sa,x0,m2):", np.array_equal(got, f1), " gpu == fma(sb,nz,m1):", np.array_equal(got, f2))
tg = torch.from_numpy(sa).to(DEV)[:,None,None,None]*x0.to(DEV) + torch.from_numpy(sb).to(DEV)[:,None,None,None]*nz.to(DEV)
print("torch-gpu eager == separate:", np.array_equal(tg.cpu().numpy(), sep))
