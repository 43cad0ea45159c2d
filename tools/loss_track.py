"""The same 12 optimizer steps (same synthetic batch order, same initial weights, lr 1e-4) on the configs[4] network in the three
arithmetic modes: fp32-equivalent tape, bf16 tape, fp16 tape + GradScaler (through Accelerator / train_loop.train_step, the
reference's loop body).  The losses must track each other: a kernel that is wrong in one mode shows here before it shows anywhere.
Usage: loss_track.py [batch] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drivescenegen_amd as d
from drivescenegen_amd import synth
from drivescenegen_amd.configs import CFG5, synth_weights
from drivescenegen_amd.train_loop import train_step

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
c = CFG5["in_channels"]
data = [torch.from_numpy(synth.synth_scene_rasters(b, c, 256, 256, 100 + i)).cuda() for i in range(4)]
out = {}
for mode in ("no", "bf16", "fp16"):
    torch.manual_seed(7)   # (train_step draws noise and timesteps from torch's generators)
    acc = d.Accelerator(mixed_precision=mode)
    net = synth_weights(d.UNet2DModel(**CFG5)).cuda()
    opt = d.AdamW(net.parameters(), lr=1e-4)
    lrs = d.get_cosine_schedule_with_warmup(optimizer=opt, num_warmup_steps=2, num_training_steps=1000)
    net, opt, lrs = acc.prepare(net, opt, lrs)
    sch = d.DDPMScheduler()
    losses = []
    for i in range(steps):
        losses.append(float(train_step(acc, net, sch, opt, lrs, data[i % len(data)])))
    out[mode] = losses
    print(mode.ljust(5), " ".join(f"{v:.4f}" for v in losses), flush=True)
ref = out["no"]
for mode in ("bf16", "fp16"):
    worst = max(abs(a - r) / r for a, r in zip(out[mode], ref))
    print(f"{mode}: largest relative deviation from the fp32-equivalent run {worst:.3e}")
    assert worst < 0.05 and out[mode][-1] < out[mode][0], mode
print("ok")
