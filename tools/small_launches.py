"""Which Python lines launch the training step's small torch kernels (fills, device-to-device copies, elementwise ops)?
One profiled step of tools/train_bench.py's loop; per (kernel, innermost drivescenegen_amd / tools frame): launches and device time.
Usage: small_launches.py [batch] [fp32|bf16|fp16]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drivescenegen_amd as d
from drivescenegen_amd import synth
from drivescenegen_amd.configs import CFG3, CFG5, synth_weights

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
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


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    dev_us = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
    if not ev.kernels and ev.name not in ("aten::copy_", "aten::fill_", "aten::zero_"):   # (plain memcpys / memsets carry no kernel record)
        continue
    frame = next((f for f in ev.stack if "drivescenegen_amd" in f or "tools/" in f), ev.stack[0] if ev.stack else "?")
    key = (ev.name, frame.strip()[-90:])
    agg[key][0] += max(1, len(ev.kernels))
    agg[key][1] += dev_us
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for _, v in rows)
print(f"{tot} torch-launched kernels in one step")
for (name, frame), (n, us) in rows[:40]:
    print(f"{n:5d} {us:9.1f} us  {name:28s} {frame}")
