"""The other BASELINE configs at full size (they are parity / capacity cases, not the bench line):
  cfg4: 512x512x4, 6 levels (64,64,128,128,256,512), attention at 32^2 and 16^2, 100-step DDIM, B=8, 1 GPU
  cfg3: the bench net's training step at B=64 (fp32-equivalent engine)
  cfg5: 256x256x8 training step at B=128 (the engine trains in fp32-equivalent, not bf16)
Usage: python tools/config_bench.py [cfg4] [cfg3] [cfg5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drivescenegen_amd as d
from drivescenegen_amd import synth
from tests.common import CFG2, synth_weights

which = sys.argv[1:] or ["cfg4", "cfg3", "cfg5"]
if "cfg4" in which:
    cfg = dict(sample_size=512, in_channels=4, out_channels=4, layers_per_block=2,
               block_out_channels=(64, 64, 128, 128, 256, 512),
               down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D",) * 2,
               up_block_types=("AttnUpBlock2D",) * 2 + ("UpBlock2D",) * 4)
    net = synth_weights(d.UNet2DModel(**cfg)).to("cuda").eval().requires_grad_(False)
    sch = d.DDIMScheduler(); sch.set_timesteps(100)
    x = torch.from_numpy(synth.normal(1, (8, 4, 512, 512))).cuda()
    ts = [int(t) for t in sch.timesteps]
    def step(i, x):
        return sch.step(net(x, ts[i]).sample, ts[i], x).prev_sample
    for i in range(3): x = step(i, x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3, 23): x = step(i, x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    assert torch.isfinite(x).all()
    print(f"cfg4 512x512x4 B=8 ({sum(p.numel() for p in net.parameters()):,} params): {dt*1e3:.1f} ms/step, {8/dt:.1f} image-steps/s, "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    del net, x
    torch.cuda.empty_cache()

def train(cfg, b, c, tag):
    torch.cuda.reset_peak_memory_stats()
    net = synth_weights(d.UNet2DModel(**cfg)).to("cuda").train()
    opt = d.AdamW(net.parameters(), lr=1e-5); sch = d.DDPMScheduler()
    x0 = torch.from_numpy(synth.synth_scene_rasters(b, c, 256, 256, 1)).cuda()
    noise = torch.from_numpy(synth.normal(2, (b, c, 256, 256))).cuda()
    t = torch.randint(0, 1000, (b,), device="cuda")
    def step():
        loss = d.mse_loss(net(sch.add_noise(x0, noise, t), t, return_dict=False)[0], noise)
        loss.backward(); d.clip_grad_norm_(net.parameters(), 1.0); opt.step(); opt.zero_grad()
        return loss
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): loss = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{tag} training B={b} C={c}: {dt*1e3:.0f} ms/step, {b/dt:.1f} images/s, loss {float(loss.detach()):.4f}, "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    del net, opt, x0, noise
    torch.cuda.empty_cache()

if "cfg3" in which:
    train(CFG2, 64, 4, "cfg3")
if "cfg5" in which:
    train(dict(CFG2, in_channels=8, out_channels=8), 128, 8, "cfg5")
