"""Data-parallel training steps through `Accelerator` (process group, rank-0 broadcast, bucketed gradient all-reduce).
Under a launcher:  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/ddp_smoke.py [cfg] [batch] [steps]
With DSG_FORCE_COLLECTIVES=1 and N = 1 every collective of the path still runs on a one-rank RCCL communicator (what
tests/test_gpu_rccl_one_rank.py does on the one-GPU box).  cfg: CFG1 (default, 919 k parameters = one bucket) or any name
of drivescenegen_amd.configs (DEFAULT3: the train.py:39-57 network, 56.6 M parameters = 8 buckets of >= 25 MB).
Prints one line per step, the bucket trace of the last step (DSG_DDP_TRACE=1) and a checksum of the parameters.
DSG_SMOKE_SHARD=1: rank r trains on rows [r B / W, (r + 1) B / W) of the batch (default: every rank on the whole batch -- the
average of W equal gradients is that gradient, bit for bit).  DSG_DIST_BACKEND=gloo lets two ranks share one GPU.
--selfcheck (implied whenever WORLD_SIZE > 1; DSG_DDP_SELFCHECK=0 turns it off): before the steps, step 0 is run twice from the
same weights -- gradient buckets all-reduced from INSIDE the backward walk (the default, DSG_DDP_OVERLAP=1) and launched AFTER it
(DSG_DDP_OVERLAP=0: nothing of RCCL beside the backward kernels) -- and loss + every parameter after the update are compared
bitwise; one line `selfcheck ...` per rank, exit code 3 at the end of the run if the two modes differ.  The first run on real
xGMI links therefore says by itself whether overlapping RCCL's reduction kernels with the backward changes a bit."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drivescenegen_amd as d
from drivescenegen_amd import configs, synth
from drivescenegen_amd.training import GradBuckets
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg_name = argv[0] if len(argv) > 0 else "CFG1"
B = int(argv[1]) if len(argv) > 1 else 4
steps = int(argv[2]) if len(argv) > 2 else 3
cfg = getattr(configs, cfg_name)
ss = cfg["sample_size"]
H, W = (ss, ss) if isinstance(ss, int) else ss
C = cfg["in_channels"]


def say(*parts):   # one write() per line: two ranks share the launcher's pipe, and print() hands a line over in pieces
    sys.stdout.flush()
    os.write(1, (" ".join(str(x) for x in parts) + "\n").encode())


acc = d.Accelerator()
sch = d.DDPMScheduler()


def fresh():
    net = configs.synth_weights(d.UNet2DModel(**cfg)).train()
    opt = d.AdamW(net.parameters(), lr=1e-4)
    return acc.prepare(net, opt)


def one_step(net, opt, x0, noise, t):
    with acc.accumulate(net):
        loss = d.mse_loss(net(sch.add_noise(x0, noise, t), t, return_dict=False)[0], noise)
        acc.backward(loss)
        acc.clip_grad_norm_(net.parameters(), 1.0)
        opt.step(); opt.zero_grad()
    return loss


def selfcheck(x0, noise, t):
    """step 0 with the buckets launched from inside the walk and after it: loss and parameters, bit for bit"""
    got = {}
    for overlap in (True, False):
        acc.ddp_overlap = overlap
        net, opt = fresh()
        loss = one_step(net, opt, x0, noise, t)
        torch.cuda.synchronize()
        b = getattr(acc, "_buckets", None)
        got[overlap] = (float(loss.detach()).hex(), [p.detach().clone() for p in net.parameters()],
                        list(b.last_launch_order) if b is not None else [], b.overlap if b is not None else None)
        del net, opt
    acc.ddp_overlap = None
    same_loss = got[True][0] == got[False][0]
    diff = max(float((a - b).abs().max()) for a, b in zip(got[True][1], got[False][1]))
    same_par = all(torch.equal(a, b) for a, b in zip(got[True][1], got[False][1]))
    ok = same_loss and same_par and got[True][2] == got[False][2] and got[True][3] is not False and got[False][3] is not True
    say("selfcheck rank", acc.process_index, "overlap_vs_deferred", "loss_equal", same_loss, "params_equal", same_par,
        "max_abs_param_diff", diff, "bucket_order_equal", got[True][2] == got[False][2], "buckets", len(got[True][2]),
        "OK" if ok else "MISMATCH")
    return ok


x0 = torch.from_numpy(synth.synth_scene_rasters(B, C, H, W, 1)).to(acc.device)
noise = torch.from_numpy(synth.normal(2, (B, C, H, W))).to(acc.device)
t = torch.tensor([3, 250, 600, 999] * ((B + 3) // 4), device=acc.device)[:B]
if os.environ.get("DSG_SMOKE_SHARD") == "1" and acc.num_processes > 1:
    per = B // acc.num_processes
    rows = slice(acc.process_index * per, (acc.process_index + 1) * per)
    x0, noise, t = x0[rows].contiguous(), noise[rows].contiguous(), t[rows].contiguous()
check_ok = True
if ("--selfcheck" in sys.argv or acc.num_processes > 1) and os.environ.get("DSG_DDP_SELFCHECK") != "0":
    check_ok = selfcheck(x0, noise, t)
net, opt = fresh()
for i in range(steps):
    loss = one_step(net, opt, x0, noise, t)
    say("rank", acc.process_index, "of", acc.num_processes, "step", i, "loss", float(loss.detach()).hex())
acc.wait_for_everyone()
b = getattr(acc, "_buckets", None)
say("collectives", "on" if torch.distributed.is_initialized() else "off",
      "backend", torch.distributed.get_backend() if torch.distributed.is_initialized() else "-",
      "buckets", len(b.buckets) if b is not None else 0)
if GradBuckets.last_trace is not None:
    say("trace", json.dumps(GradBuckets.last_trace))
say("checksum", float(sum(p.detach().double().abs().sum() for p in net.parameters())).hex())
if not check_ok:
    sys.exit(3)
