"""Training-step throughput on one GPU (BASELINE configs[2] shape: 256x256x4 default U-Net, DDPM step):
add_noise -> fwd -> mse -> bwd -> clip -> AdamW.  images/s and ms/step.  Usage: train_bench.py [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drivescenegen_amd as d
from drivescenegen_amd import synth
from tests.common import CFG2, synth_weights

b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
net = synth_weights(d.UNet2DModel(**CFG2)).to("cuda").train()
opt = d.AdamW(net.parameters(), lr=1e-5)
sch = d.DDPMScheduler()
x0 = torch.from_numpy(synth.synth_scene_rasters(b, 4, 256, 256, 1)).cuda()
noise = torch.from_numpy(synth.normal(2, (b, 4, 256, 256))).cuda()
t = torch.randint(0, 1000, (b,), device="cuda")


def step():
    noisy = sch.add_noise(x0, noise, t)
    loss = d.mse_loss(net(noisy, t, return_dict=False)[0], noise)
    loss.backward()
    d.clip_grad_norm_(net.parameters(), 1.0)
    opt.step()
    opt.zero_grad()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"batch {b}: {dt*1e3:.1f} ms/step, {b/dt:.1f} images/s, loss {float(loss.detach()):.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, "
      f"~{3*352.98e9*b/dt/1e12:.1f} TF/s (3x fwd FLOPs)")
