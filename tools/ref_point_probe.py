"""Where the reference's own training step (train.py: 3-channel default U-Net, fp16 AMP + GradScaler, batch 14) spends its time
beyond the tape: the same loop with one ingredient removed at a time.
Usage: ref_point_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drivescenegen_amd as d
from drivescenegen_amd import synth, training
from drivescenegen_amd.configs import DEFAULT3, synth_weights
from drivescenegen_amd.train_loop import train_step, train_steps

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
batch = 14


def build():
    acc = d.Accelerator(mixed_precision="fp16")
    net = synth_weights(d.UNet2DModel(**DEFAULT3)).to(dev)
    opt = d.AdamW(net.parameters(), lr=1e-4)
    lrs = d.get_cosine_schedule_with_warmup(optimizer=opt, num_warmup_steps=500, num_training_steps=50000)
    net, opt, lrs = acc.prepare(net, opt, lrs)
    return acc, net, opt, lrs


def timed(fn, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    print(f"{label:60s} {(time.perf_counter() - t0) / steps * 1e3:7.2f} ms/step", flush=True)


sch = d.DDPMScheduler()
x0 = torch.from_numpy(synth.synth_scene_rasters(batch, 3, 256, 256, 14555)).to(dev)
acc, net, opt, lrs = build()
timed(lambda: train_step(acc, net, sch, opt, lrs, x0), "serial loop (host noise drawn on the critical path)")
it = {"g": None}


def ahead_step():
    for _ in train_steps(acc, net, sch, opt, lrs, [x0, x0], True):
        pass


t0 = None
for _ in range(2):
    ahead_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps // 2):
    ahead_step()
torch.cuda.synchronize()
print(f"{'noise drawn one step ahead (pairs of steps)':60s} {(time.perf_counter() - t0) / (steps // 2 * 2) * 1e3:7.2f} ms/step", flush=True)
noise_dev = torch.randn(x0.shape, device=dev)
timed(lambda: train_step(acc, net, sch, opt, lrs, x0, noise=noise_dev), "noise resident on the device (no draw, no H2D copy)")
orig_step = training.GradScaler.step


def nosync_step(self, optimizer, params):
    self.unscale_(params)
    optimizer.step()
    return False


training.GradScaler.step = nosync_step
timed(lambda: train_step(acc, net, sch, opt, lrs, x0, noise=noise_dev), "... and no found-inf read-back before the optimizer step")
training.GradScaler.step = orig_step
acc2 = d.Accelerator(mixed_precision="fp16")
acc2.scaler = None   # (fp16 tape without the GradScaler: no loss scaling, no unscale pass)
net.set_compute_dtype("fp16")
timed(lambda: train_step(acc2, net, sch, opt._PreparedOptimizer__dummy if False else d.training._PreparedOptimizer(opt.optimizer, acc2), lrs, x0, noise=noise_dev),
      "... and no GradScaler at all (no loss scaling / unscale pass)")


def bare():
    t = torch.randint(0, 1000, (batch,), device=dev)
    noisy = sch.add_noise(x0, noise_dev, t)
    loss = d.mse_loss(net(noisy, t, return_dict=False)[0], noise_dev)
    loss.backward()
    d.clip_grad_norm_(net.parameters(), 1.0)
    opt.optimizer.step()
    opt.optimizer.zero_grad()


timed(bare, "bare tape (tools/train_bench.py's loop)")
