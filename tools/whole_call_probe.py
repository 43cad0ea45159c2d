"""Where the wall time of DDPMPipeline.__call__ goes at batch 1 (training_pipeline.py:26-32's evaluate call): the same 750 steps
with the device RNG, with the seeded CPU generator (the reference's call), and the bare step loop bench.py times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drivescenegen_amd as d
from drivescenegen_amd import synth
from drivescenegen_amd.configs import DEFAULT3, synth_weights
dev = torch.device("cuda", 0)
net = synth_weights(d.UNet2DModel(**DEFAULT3)).to(dev).eval().requires_grad_(False)
pipe = d.DDPMPipeline(unet=net, scheduler=d.DDPMScheduler())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 750
pipe(batch_size=1, num_inference_steps=20)
pipe(batch_size=1, num_inference_steps=20, generator=torch.manual_seed(1))
torch.cuda.synchronize()


def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return time.perf_counter() - t0


sch = d.DDPMScheduler(); sch.set_timesteps(n)
x = torch.from_numpy(synth.normal(1, (1, 3, 256, 256))).to(dev)
nz = torch.from_numpy(synth.normal(2, (1, 3, 256, 256))).to(dev)
ts = [int(t) for t in sch.timesteps]


def bare():
    global x
    for t in ts:
        x = sch.step(net(x, t).sample, t, x, variance_noise=nz).prev_sample


for rep in range(2):
    tb = timed(bare)
    t0 = timed(lambda: pipe(batch_size=1, num_inference_steps=n, output_type="np.array"))
    t1 = timed(lambda: pipe(batch_size=1, num_inference_steps=n, generator=torch.manual_seed(14555), output_type="np.array"))
    print(f"steps {n}: bare loop {tb / n * 1e3:.3f} ms/step | pipeline, device RNG {t0 / n * 1e3:.3f} ({t0 / tb:.3f}x) | "
          f"pipeline, CPU generator {t1 / n * 1e3:.3f} ({t1 / tb:.3f}x)")

# --- alternatives for getting a CPU generator's per-step noise to the GPU (all draw on a worker thread, one tensor ahead) ---
import threading
import ctypes as C
from drivescenegen_amd import _lib
lib = _lib.load()
shape = (1, 3, 256, 256)


def loop(mode):
    gen = torch.manual_seed(14555)
    x = torch.randn(shape, generator=gen).to(dev)
    pinned = [torch.empty(shape).pin_memory() for _ in range(2)]
    onchip = [torch.empty(shape, device=dev) for _ in range(2)]
    used = [None, None]
    main = torch.cuda.current_stream(dev)
    th = None

    def start(k):
        nonlocal th
        i = k & 1
        if used[i] is not None:
            used[i].synchronize()
        th = threading.Thread(target=lambda: torch.randn(shape, generator=gen, out=pinned[i]), daemon=True)
        th.start()
    start(1)
    for k, t in enumerate(ts):
        eps = net(x, t).sample
        if t > 0:
            i = (k + 1) & 1
            th.join()
            s = sch.step_scalars(t)
            prev = torch.empty_like(x)
            if mode == "main":
                onchip[i].copy_(pinned[i], non_blocking=True)
                nptr = onchip[i].data_ptr()
            else:   # zero-copy: the step kernel reads the pinned host buffer over PCIe
                nptr = pinned[i].data_ptr()
            _lib.check(lib.dsg_ddpm_step(_lib.ptr(x), _lib.ptr(eps), nptr, _lib.ptr(prev), x.numel(), s["sqrt_beta_prod_t"],
                                         s["sqrt_alpha_prod_t"], 1.0, s["coef_x0"], s["coef_xt"], s["sigma"], _lib.stream_ptr(dev)))
            used[i] = torch.cuda.Event()
            used[i].record(main)
            x = prev
            if k + 1 < len(ts) and ts[k + 1] > 0:
                start(k + 2)
        else:
            x = sch.step(eps, t, x).prev_sample
    return x


ref = pipe(batch_size=1, num_inference_steps=n, generator=torch.manual_seed(14555), output_type="np.array").images
for mode in ("main", "zerocopy", "main", "zerocopy"):
    out = {}
    t = timed(lambda: out.setdefault("x", loop(mode)))
    img = (out["x"] / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy()
    import numpy as np
    print(f"{mode}: {t / n * 1e3:.3f} ms/step, same image as the pipeline: {np.array_equal(img, ref)}")
