"""Three data-parallel training steps through `Accelerator` (process group, rank-0 broadcast, bucketed gradient
all-reduce).  Under a launcher:  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/ddp_smoke.py
With DSG_FORCE_COLLECTIVES=1 and N = 1 every collective of the path still runs on a one-rank RCCL communicator (what
tests/test_gpu_rccl_one_rank.py does on the one-GPU box).  Prints one line per step and a checksum of the parameters."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drivescenegen_amd as d
from drivescenegen_amd import configs, synth
acc = d.Accelerator()
net = configs.synth_weights(d.UNet2DModel(**configs.CFG1)).train()
opt = d.AdamW(net.parameters(), lr=1e-4)
net, opt = acc.prepare(net, opt)
sch = d.DDPMScheduler()
x0 = torch.from_numpy(synth.synth_scene_rasters(4, 3, 64, 64, 1)).to(acc.device)
noise = torch.from_numpy(synth.normal(2, (4, 3, 64, 64))).to(acc.device)
t = torch.tensor([3, 250, 600, 999], device=acc.device)
for i in range(3):
    with acc.accumulate(net):
        loss = d.mse_loss(net(sch.add_noise(x0, noise, t), t, return_dict=False)[0], noise)
        acc.backward(loss)
        acc.clip_grad_norm_(net.parameters(), 1.0)
        opt.step(); opt.zero_grad()
    print("rank", acc.process_index, "of", acc.num_processes, "step", i, "loss", float(loss.detach()).hex())
acc.wait_for_everyone()
b = getattr(acc, "_buckets", None)
print("collectives", "on" if torch.distributed.is_initialized() else "off",
      "backend", torch.distributed.get_backend() if torch.distributed.is_initialized() else "-",
      "buckets", len(b.buckets) if b is not None else 0)
print("checksum", float(sum(p.detach().double().abs().sum() for p in net.parameters())).hex())
