"""Data-parallel training steps through `Accelerator` (process group, rank-0 broadcast, bucketed gradient all-reduce).
Under a launcher:  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/ddp_smoke.py [cfg] [batch] [steps]
With DSG_FORCE_COLLECTIVES=1 and N = 1 every collective of the path still runs on a one-rank RCCL communicator (what
tests/test_gpu_rccl_one_rank.py does on the one-GPU box).  cfg: CFG1 (default, 919 k parameters = one bucket) or any name
of drivescenegen_amd.configs (DEFAULT3: the train.py:39-57 network, 56.6 M parameters = 8 buckets of >= 25 MB).
Prints one line per step, the bucket trace of the last step (DSG_DDP_TRACE=1) and a checksum of the parameters.
DSG_SMOKE_SHARD=1: rank r trains on rows [r B / W, (r + 1) B / W) of the batch (default: every rank on the whole batch -- the
average of W equal gradients is that gradient, bit for bit).  DSG_DIST_BACKEND=gloo lets two ranks share one GPU."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drivescenegen_amd as d
from drivescenegen_amd import configs, synth
from drivescenegen_amd.training import GradBuckets
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "CFG1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = getattr(configs, cfg_name)
ss = cfg["sample_size"]
H, W = (ss, ss) if isinstance(ss, int) else ss
C = cfg["in_channels"]


def say(*parts):   # one write() per line: two ranks share the launcher's pipe, and print() hands a line over in pieces
    sys.stdout.flush()
    os.write(1, (" ".join(str(x) for x in parts) + "\n").encode())


acc = d.Accelerator()
net = configs.synth_weights(d.UNet2DModel(**cfg)).train()
opt = d.AdamW(net.parameters(), lr=1e-4)
net, opt = acc.prepare(net, opt)
sch = d.DDPMScheduler()
x0 = torch.from_numpy(synth.synth_scene_rasters(B, C, H, W, 1)).to(acc.device)
noise = torch.from_numpy(synth.normal(2, (B, C, H, W))).to(acc.device)
t = torch.tensor([3, 250, 600, 999] * ((B + 3) // 4), device=acc.device)[:B]
if os.environ.get("DSG_SMOKE_SHARD") == "1" and acc.num_processes > 1:
    per = B // acc.num_processes
    rows = slice(acc.process_index * per, (acc.process_index + 1) * per)
    x0, noise, t = x0[rows].contiguous(), noise[rows].contiguous(), t[rows].contiguous()
for i in range(steps):
    with acc.accumulate(net):
        loss = d.mse_loss(net(sch.add_noise(x0, noise, t), t, return_dict=False)[0], noise)
        acc.backward(loss)
        acc.clip_grad_norm_(net.parameters(), 1.0)
        opt.step(); opt.zero_grad()
    say("rank", acc.process_index, "of", acc.num_processes, "step", i, "loss", float(loss.detach()).hex())
acc.wait_for_everyone()
b = getattr(acc, "_buckets", None)
say("collectives", "on" if torch.distributed.is_initialized() else "off",
      "backend", torch.distributed.get_backend() if torch.distributed.is_initialized() else "-",
      "buckets", len(b.buckets) if b is not None else 0)
if GradBuckets.last_trace is not None:
    say("trace", json.dumps(GradBuckets.last_trace))
say("checksum", float(sum(p.detach().double().abs().sum() for p in net.parameters())).hex())
