"""Forward (denoising-step) throughput of one network / precision on one GPU, without bench.py's extra legs.
Usage: fwd_bench.py [cfg2|cfg4|cfg5|default3] [batch] [steps] [fp32|bf16|fp16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drivescenegen_amd as d
from drivescenegen_amd import configs, synth

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
cfg = {"cfg2": configs.CFG2, "cfg4": configs.CFG4, "cfg5": configs.CFG5, "default3": configs.DEFAULT3}[name]
ss = cfg["sample_size"]
h, w = (ss, ss) if isinstance(ss, int) else ss
net = configs.synth_weights(d.UNet2DModel(**cfg)).to("cuda").eval().requires_grad_(False).set_compute_dtype(dtype)
sch = d.DDIMScheduler()
sch.set_timesteps(50)
x = torch.from_numpy(synth.normal(1, (b, cfg["in_channels"], h, w))).cuda()
ts = [int(t) for t in sch.timesteps]
for i in range(5):
    x = sch.step(net(x, ts[i]).sample, ts[i], x).prev_sample
torch.cuda.synchronize()
dump = os.environ.get("PROF_DUMP")   # per-launch HIP-event records of the conv kernels (csv)
if dump:
    from drivescenegen_amd import _lib
    _lib.load().dsg_prof_enable(1)
t0 = time.perf_counter()
for i in range(steps):
    t = ts[(5 + i) % 50]
    x = sch.step(net(x, t).sample, t, x).prev_sample
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{name} {dtype} batch {b}: {dt*1e3:.2f} ms/step, {b/dt:.1f} image-steps/s")
if dump:
    _lib.check(_lib.load().dsg_prof_dump(dump.encode()))
