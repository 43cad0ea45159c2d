import os, sys, torch
sys.path.insert(0, "/root/repo")
import drivescenegen_amd as d
from drivescenegen_amd import synth
from tests.common import CFG1, synth_weights
acc = d.Accelerator()
net = synth_weights(d.UNet2DModel(**CFG1)).train()
opt = d.AdamW(net.parameters(), lr=1e-4)
net, opt = acc.prepare(net, opt)
sch = d.DDPMScheduler()
x0 = torch.from_numpy(synth.synth_scene_rasters(4, 3, 64, 64, 1)).to(acc.device)
noise = torch.from_numpy(synth.normal(2, (4, 3, 64, 64))).to(acc.device)
t = torch.randint(0, 1000, (4,), device=acc.device)
for i in range(3):
    with acc.accumulate(net):
        loss = d.mse_loss(net(sch.add_noise(x0, noise, t), t, return_dict=False)[0], noise)
        acc.backward(loss)
        acc.clip_grad_norm_(net.parameters(), 1.0)
        opt.step(); opt.zero_grad()
    print("rank", acc.process_index, "of", acc.num_processes, "step", i, "loss", float(loss.detach()))
