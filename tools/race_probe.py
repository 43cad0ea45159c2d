"""Determinism under GPU sharing: prints checksums of (a) inference forwards and (b) training steps; run several copies at once
(tools/race_probe.sh) and compare with a solo run.  A kernel with a latent race (a missing barrier, a consumer ahead of its
producer) is deterministic when it has the CU to itself and changes bits when another process's waves perturb the timing.
Usage: race_probe.py [fwd|train] [cfg] [batch] [dtype] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drivescenegen_amd as d
from drivescenegen_amd import configs, synth

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
cfg_name = sys.argv[2] if len(sys.argv) > 2 else "CFG1"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dtype = sys.argv[4] if len(sys.argv) > 4 else "fp32"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
cfg = getattr(configs, cfg_name)
ss = cfg["sample_size"]
H, W = (ss, ss) if isinstance(ss, int) else ss
C = cfg["in_channels"]
dev = torch.device("cuda")
x0 = torch.from_numpy(synth.synth_scene_rasters(B, C, H, W, 1)).to(dev)
noise = torch.from_numpy(synth.normal(2, (B, C, H, W))).to(dev)
t = torch.tensor([3, 250, 600, 999] * ((B + 3) // 4), device=dev)[:B]
sch = d.DDPMScheduler()
out = []
if what == "fwd":
    net = configs.synth_weights(d.UNet2DModel(**cfg)).to(dev).eval().requires_grad_(False).set_compute_dtype(dtype)
    x = sch.add_noise(x0, noise, t)
    for i in range(reps):
        y = net(x, t).sample
        out.append(float(y.double().abs().sum()).hex())
elif what == "tfwd":   # the TRAINING forward alone (the tape is built, never walked): same parameters every repetition
    net = configs.synth_weights(d.UNet2DModel(**cfg)).to(dev).train().set_compute_dtype(dtype)
    x = sch.add_noise(x0, noise, t)
    for i in range(reps):
        y = net(x, t, return_dict=False)[0]
        out.append(float(y.detach().double().abs().sum()).hex()[4:16])
else:
    net = configs.synth_weights(d.UNet2DModel(**cfg)).to(dev).train().set_compute_dtype(dtype)
    opt = d.AdamW(net.parameters(), lr=1e-4)
    for i in range(reps):
        loss = d.mse_loss(net(sch.add_noise(x0, noise, t), t, return_dict=False)[0], noise)
        loss.backward()
        cs = lambda ts: float(sum((t.double() * 1.000001).abs().sum() for t in ts)).hex()[4:14]
        gsum = cs([p.grad for p in net.parameters() if p.grad is not None])
        d.clip_grad_norm_(net.parameters(), 1.0)
        csum = cs([p.grad for p in net.parameters() if p.grad is not None])
        opt.step()
        psum = cs([p.detach() for p in net.parameters()])
        opt.zero_grad()
        out.append(float(loss.detach()).hex()[4:12] + "/g" + gsum + "/c" + csum + "/p" + psum)
torch.cuda.synchronize()
if os.environ.get("RACE_DISTINCT") == "1":   # (under a background load: how many repetitions disagree with the most common result)
    from collections import Counter
    cnt = Counter(out)
    top = cnt.most_common(1)[0]
    os.write(1, f"{what} {cfg_name} {B} {dtype}: {len(out)} repetitions, {len(cnt)} distinct results, {len(out) - top[1]} off the most common ({top[0]})\n".encode())
else:
    os.write(1, (" ".join([what, cfg_name, str(B), dtype] + out) + "\n").encode())
