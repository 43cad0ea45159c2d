"""Training-step throughput on one GPU: add_noise -> fwd -> mse -> bwd -> clip -> AdamW; images/s and ms/step.
Usage: train_bench.py [batch] [steps] [fp32|bf16|fp16]   (fp32: BASELINE configs[2] network, 256x256x4;
bf16 / fp16: configs[4] network, 256x256x8, mixed-precision tape)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drivescenegen_amd as d
from drivescenegen_amd import synth
from drivescenegen_amd.configs import CFG3, CFG5, synth_weights

b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dtype = sys.argv[3] if len(sys.argv) > 3 else "fp32"
cfg = CFG3 if dtype == "fp32" else CFG5
c = cfg["in_channels"]
net = synth_weights(d.UNet2DModel(**cfg)).to("cuda").train().set_compute_dtype(dtype)
opt = d.AdamW(net.parameters(), lr=1e-5)
sch = d.DDPMScheduler()
x0 = torch.from_numpy(synth.synth_scene_rasters(b, c, 256, 256, 1)).cuda()
noise = torch.from_numpy(synth.normal(2, (b, c, 256, 256))).cuda()
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
dump = os.environ.get("PROF_DUMP")   # per-launch HIP-event records of the conv / weight-gradient kernels (csv)
if dump:
    from drivescenegen_amd import _lib
    _lib.load().dsg_prof_enable(1)
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{dtype} batch {b}: {dt*1e3:.1f} ms/step, {b/dt:.1f} images/s, loss {float(loss.detach()):.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, "
      f"~{3*352.98e9*b/dt/1e12:.1f} TF/s (3x fwd FLOPs)")
if dump:
    _lib.check(_lib.load().dsg_prof_dump(dump.encode()))
