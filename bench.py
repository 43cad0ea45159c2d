#!/usr/bin/env python3
"""Headline benchmark: denoising image-steps/sec of the U-Net forward (+ scheduler step) on
256x256 BEV rasters -- BASELINE.json configs[1]: 256x256x4 raster, DriveSceneGen default U-Net
(train.py:39-57 with 4 in/out channels), 50-step DDIM, batch 16 per GPU, fp32.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one denoising step of the whole batch: dsg_unet_forward + dsg_ddim_step through the C ABI,
inputs already resident in HBM.  Samples are independent, so ranks shard them with NO data-path
collective (weak scaling: 16 samples per GPU); the only communication is the timing barrier / max.
Rank 0 prints ONE JSON line, with
  roofline     -- the dominant kernel (3x3 stride-1 implicit-GEMM conv, fp32-equivalent as an fp16x2 split on the f16
                  matrix cores): algorithmic FLOPs of its launches / their summed duration, from HIP events
                  recorded on the launch stream inside the timed region (dsg_prof_*), against 2500 / 3 = 833 TF/s
                  (the guide's dense f16 MFMA peak over the split's three products per MAC);
  cpu_baseline -- the torch-CPU oracle (oracle/, kind "port") timed on the host cores on a bounded
                  sample (a few steps at batch 2) of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("DSG_TESTING", "1")   # --pure-f32 / --separate-gn-stats flip kernel-selection switches (a test hook)

import torch  # noqa: E402

PEAK_F32_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: f32 vector = f32 MFMA peak
PEAK_F16_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak (same guide)
PEAK_HBM_GBS = 8000.0
CFG4_FLOPS_IMG = 531e9   # configs[3] forward, per image


def _cfg():
    from drivescenegen_amd.configs import CFG2
    return CFG2


def _local_device():
    """one rank per GPU: LOCAL_RANK; more local ranks than GPUs (the two-ranks-on-one-GPU test) wrap around"""
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    lr = int(os.environ.get("LOCAL_RANK", 0))
    return lr % n if n else lr


def gpu_leg(args, rank, world):
    import drivescenegen_amd as d
    from drivescenegen_amd import _lib, synth
    from drivescenegen_amd.configs import synth_weights

    cfg = _cfg()
    dev = torch.device("cuda", _local_device())
    torch.cuda.set_device(dev)
    net = synth_weights(d.UNet2DModel(**cfg)).to(dev).eval().requires_grad_(False)
    sch = d.DDIMScheduler()
    sch.set_timesteps(args.ddim_steps)
    ts = [int(t) for t in sch.timesteps]
    b = args.batch
    # x_T ~ N(0,1): each rank draws its own rows of the global batch (stream = rank)
    x = torch.from_numpy(synth.normal(14555, (b, cfg["in_channels"], 256, 256), stream=7 + rank)).to(dev)
    lib = _lib.load()
    if args.pure_f32:
        _lib.check(lib.dsg_set_tuning(2, 0))
    if args.separate_gn_stats:
        _lib.check(lib.dsg_set_tuning(5, 0))

    def step(i, x):
        t = ts[i % len(ts)]
        eps = net(x, t).sample
        return sch.step(eps, t, x).prev_sample

    def barrier():
        torch.cuda.synchronize(dev)
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # untimed: the W warm-up steps, preceded by as many more as it takes to make 10 -- the first process on a freshly
    # booted box needs ~0.2 s of load before the GPU reaches its steady clocks (3 steps: 1.5 % low; 12: same as later runs)
    for i in range(max(0, 10 - args.warmup)):
        x = step(i, x)
    for i in range(args.warmup):
        x = step(i, x)
    barrier()
    # HIP events bracket the conv launches of every PROF_EVERY-th step of the timed region (all 50 DDIM timesteps run
    # the same launches): the roofline figures are live, from inside the timed region, at ~1/5 of the event cost
    lib.dsg_prof_enable(1 if not args.no_prof else 0)
    clock = StepClock(dev)
    t0 = time.perf_counter()
    clock.tick()
    for i in range(args.steps):
        if not args.no_prof:
            lib.dsg_prof_enable(3 if i % PROF_EVERY == 0 else 2)
        x = step(args.warmup + i, x)
        clock.tick()
    barrier()
    dt = time.perf_counter() - t0
    per_rank = [dt]
    if torch.distributed.is_initialized():
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(tt) for _ in range(torch.distributed.get_world_size())]
        torch.distributed.all_gather(every, tt)     # (each rank's own wall time: the spread shows a slow GPU / link)
        per_rank = [float(v.item()) for v in every]
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(x).all()

    prof = {}
    if not args.no_prof:
        names = {0: "conv3x3_s1_mfma_f32", 1: "conv3x3_upsample_mfma_f32", 2: "conv3x3_s2_mfma_f32",
                 3: "conv1x1_mfma_f32", 4: "conv_direct_valu", 6: "conv3x3_s1_mfma_f16x2split",
                 7: "conv3x3_upsample_mfma_f16x2split", 8: "conv1x1_mfma_f16x2split",
                 10: "conv3x3_s1_mfma_f16x2split_two_wg_per_cu", 11: "conv_in_image_to_blocked",
                 12: "conv_out_blocked_to_image", 13: "conv3x3_plus_fused_shortcut_f16x2split",
                 14: "conv3x3_plus_fused_shortcut_f16x2split_two_wg_per_cu"}
        for kid, nm in names.items():
            ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
            _lib.check(lib.dsg_prof_summary(kid, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)))
            if n.value:
                prof[nm] = dict(launches=n.value, total_ms=ms.value, avg_ms=ms.value / n.value,
                                tflops=fl.value / (ms.value * 1e-3) / 1e12,
                                alg_gbs=by.value / (ms.value * 1e-3) / 1e9,
                                flops_per_launch=fl.value / n.value)
        if args.prof_dump and rank == 0:
            _lib.check(lib.dsg_prof_dump(args.prof_dump.encode()))
        lib.dsg_prof_enable(0)
    return dt, prof, per_rank, clock.spread()


def cpu_leg(args):
    """The oracle (torch-CPU restatement, kind 'port') on all host cores: 1 warm-up + 2 timed
    denoising steps at batch 2 of the same network / raster size."""
    from oracle.scheduler_oracle import OracleDDIMScheduler
    from oracle.unet_oracle import OracleUNet2DModel
    from drivescenegen_amd import synth
    from drivescenegen_amd.configs import synth_weights

    cores = max(1, min(os.cpu_count() or 1, args.cpu_threads))
    torch.set_num_threads(cores)
    cfg = _cfg()
    net = synth_weights(OracleUNet2DModel(**cfg)).eval()
    sch = OracleDDIMScheduler()
    sch.set_timesteps(args.ddim_steps)
    bs, nsteps = 2, 4
    x = torch.from_numpy(synth.normal(14555, (bs, cfg["in_channels"], 256, 256), stream=7))
    with torch.no_grad():
        t = sch.timesteps[0]
        x = sch.step(net(x, t).sample, t, x).prev_sample  # warm-up
        t0 = time.perf_counter()
        for i in range(nsteps):
            t = sch.timesteps[1 + i]
            x = sch.step(net(x, t).sample, t, x).prev_sample
        dt = time.perf_counter() - t0
    return dict(value=bs * nsteps / dt, unit="image-steps/s", cores=cores, kind="port",
                sample=f"{nsteps} DDIM steps at batch {bs} (after 1 warm-up) of the same 256x256x4 default U-Net, "
                       f"torch-CPU fp32 oracle, {cores} threads, {dt:.1f} s")


PROF_EVERY = 5


class StepClock:
    """GPU time of each step of a timed loop from events recorded on the launch stream at the step boundaries (no host
    synchronisation inside the loop); `spread()` = min / median / max ms over the steps -- box noise made visible."""

    def __init__(self, dev):
        self.dev, self.ev = dev, []

    def tick(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(self.dev))
        self.ev.append(e)

    def spread(self):
        if len(self.ev) < 2:
            return None
        ms = sorted(a.elapsed_time(b) for a, b in zip(self.ev[:-1], self.ev[1:]))
        return {"min": ms[0], "median": ms[len(ms) // 2], "max": ms[-1], "n": len(ms)}


def _prof_rows(lib, _lib, names):
    rows = {}
    for kid, nm in names.items():
        ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        _lib.check(lib.dsg_prof_summary(kid, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)))
        if n.value:
            rows[nm] = dict(launches=n.value, total_ms=ms.value, avg_ms=ms.value / n.value,
                            tflops=fl.value / (ms.value * 1e-3) / 1e12, alg_gbs=by.value / (ms.value * 1e-3) / 1e9,
                            flops_per_launch=fl.value / n.value, bytes_per_launch=by.value / n.value)
    return rows


FWD_CLASSES_F32 = {6: "conv3x3_s1_mfma_f16x2split", 10: "conv3x3_s1_mfma_f16x2split_two_wg_per_cu",
                   13: "conv3x3_plus_fused_shortcut_f16x2split", 14: "conv3x3_plus_fused_shortcut_f16x2split_two_wg_per_cu",
                   7: "conv3x3_upsample_mfma_f16x2split", 2: "conv3x3_s2_mfma_f16x2split", 8: "conv1x1_mfma_f16x2split",
                   0: "conv3x3_s1_mfma_f32", 4: "conv_direct_valu", 11: "conv_in_image_to_blocked", 12: "conv_out_blocked_to_image"}
FWD_CLASSES_16 = {26: "conv3x3_s1_mfma_16bit", 30: "conv3x3_s1_mfma_16bit_two_wg_per_cu", 33: "conv3x3_plus_fused_shortcut_16bit",
                  34: "conv3x3_plus_fused_shortcut_16bit_two_wg_per_cu", 27: "conv3x3_upsample_mfma_16bit", 22: "conv3x3_s2_mfma_16bit",
                  28: "conv1x1_mfma_16bit", 0: "conv3x3_s1_mfma_f32", 4: "conv_direct_valu", 11: "conv_in_image_to_blocked",
                  12: "conv_out_blocked_to_image"}


# HIP-event classes 26 / 33 hold every 16-bit 3x3 stride-1 instantiation (conv_h2_launch.h: the two-workgroup classes are
# fp32-equivalent only) -- 128-cout workgroups at one per CU AND 64-cout workgroups at two: the counter pattern takes both
MIXED_PMC_PATTERNS = {"conv3x3_s1_mfma_16bit": r"conv_h2_kernel<0, [24], 3, [02], 4, [12], 3, (64|128), 1, \d, 0",
                      "conv3x3_plus_fused_shortcut_16bit": r"conv_h2_kernel<0, [24], 3, [02], 4, [12], 3, (64|128), 1, \d, 1"}
CFG4_PMC_PATTERNS = {"conv3x3_s1_mfma_f16x2split": r"conv_h2_kernel<0, 4, 3, 2, 4, 1, 3, 64, 0, 0, 0",
                     "conv3x3_s1_mfma_f16x2split_two_wg_per_cu": r"conv_h2_kernel<0, 2, 3, 2, 4, 2, 3, 64, 0, 1, 0",
                     "conv3x3_plus_fused_shortcut_f16x2split": r"conv_h2_kernel<0, 4, 3, 2, 4, 1, 3, 64, 0, 0, 1",
                     "conv3x3_plus_fused_shortcut_f16x2split_two_wg_per_cu": r"conv_h2_kernel<0, 2, 3, 2, 4, 2, 3, 64, 0, 1, 1"}

# what a reader needs next to a class's figures
CLASS_NOTES = {
    "conv3x3_upsample_mfma_f16x2split": "algorithmic FLOPs are the reference op's (3x3 conv on the nearest-x2 up-sampled map); the "
                                         "folded kernel contracts four 2x2 phase filters on the low-resolution map = 4/9 of them, "
                                         "so tflops / frac of this class read 2.25x what the matrix cores issue",
    "conv3x3_upsample_mfma_16bit": "as conv3x3_upsample_mfma_f16x2split: FLOPs counted on the up-sampled map, 4/9 of them issued",
}


def class_roofline(row, peak_tflops, step_ms, sampled_steps=1):
    """One conv class against BOTH roofs: time at the matrix-core peak for its algorithmic FLOPs, time at 8 TB/s for its
    algorithmic bytes; `bound` is the longer of the two, `frac` = that time / the measured launch time."""
    t_mfma = row["flops_per_launch"] / (peak_tflops * 1e12)
    t_hbm = row["bytes_per_launch"] / (PEAK_HBM_GBS * 1e9)
    bound = "hbm" if t_hbm > t_mfma else "mfma"
    return dict(bound=bound, achieved=row["alg_gbs"] if bound == "hbm" else row["tflops"],
                peak=PEAK_HBM_GBS if bound == "hbm" else peak_tflops, unit="GB/s" if bound == "hbm" else "TFLOP/s",
                frac=max(t_mfma, t_hbm) / (row["avg_ms"] * 1e-3), mfma_tflops=row["tflops"], mfma_frac=row["tflops"] / peak_tflops,
                alg_gbs=row["alg_gbs"], hbm_frac=row["alg_gbs"] / PEAK_HBM_GBS, avg_launch_ms=row["avg_ms"],
                launches=row["launches"], time_share=row["total_ms"] / (sampled_steps * step_ms))


def forward_leg(args, cfg, dtype, batch, steps, ddim_steps, workload, flops_img, pmc_suffix, pmc_patterns):
    """Denoising steps (dsg_unet_forward + dsg_ddim_step) of one BASELINE network / precision / batch on one GPU, with the
    HIP-event class records of every PROF_EVERY-th step and a roofline object for the class that takes the most time."""
    import drivescenegen_amd as d
    from drivescenegen_amd import _lib, synth
    from drivescenegen_amd.configs import synth_weights
    dev = torch.device("cuda", _local_device())
    net = synth_weights(d.UNet2DModel(**cfg)).to(dev).eval().requires_grad_(False).set_compute_dtype(dtype)
    sch = d.DDIMScheduler()
    sch.set_timesteps(ddim_steps)
    ts = [int(t) for t in sch.timesteps]
    ss = cfg["sample_size"]
    h, w = (ss, ss) if isinstance(ss, int) else ss
    x = torch.from_numpy(synth.normal(14555, (batch, cfg["in_channels"], h, w), stream=11)).to(dev)
    lib = _lib.load()

    def step(i, x):
        t = ts[i % len(ts)]
        return sch.step(net(x, t).sample, t, x).prev_sample
    for i in range(4):
        x = step(i, x)
    torch.cuda.synchronize(dev)
    lib.dsg_prof_enable(1)
    clock = StepClock(dev)
    t0 = time.perf_counter()
    clock.tick()
    for i in range(steps):
        lib.dsg_prof_enable(3 if i % PROF_EVERY == 0 else 2)
        x = step(4 + i, x)
        clock.tick()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    assert torch.isfinite(x).all()
    rows = _prof_rows(lib, _lib, FWD_CLASSES_F32 if dtype == "fp32" else FWD_CLASSES_16)
    lib.dsg_prof_enable(0)
    del net
    torch.cuda.empty_cache()
    step_ms = dt / steps * 1e3
    rec = {"metric": "denoising-steps/sec (U-Net fwd)", "value": batch * steps / dt, "unit": "image-steps/s",
           "ms_per_step": step_ms, "step_ms_spread": clock.spread(), "steps": steps, "dtype": dtype,
           "config": {"workload": workload, "batch": batch},
           "whole_net_tflops": batch * steps / dt * flops_img / 1e12,
           "alg_hbm_gbs_whole_step": None, "kernels": rows,
           "class_notes": {k: v for k, v in CLASS_NOTES.items() if k in rows}}
    peak = PEAK_F16_TFLOPS / (3.0 if dtype == "fp32" else 1.0)
    sampled = len(range(0, steps, PROF_EVERY))   # steps whose launches carry HIP-event records
    conv_rows = {k: v for k, v in rows.items() if k.startswith("conv3x3_s1") or k.startswith("conv3x3_plus")}
    if conv_rows:
        # every 3x3 stride-1 class priced the same way; the headline object is the class with the largest share of the step
        per_class = {k: dict(class_roofline(v, peak, step_ms, sampled), **pmc_class_traffic(pmc_patterns.get(k, r"$^"), pmc_suffix))
                     for k, v in conv_rows.items()}
        dom = max(per_class, key=lambda k: per_class[k]["time_share"])
        rec["roofline"] = dict(per_class[dom], kernel_class=dom, peak_note=(
            "2500 TF/s dense f16 MFMA / 3 products per fp32-equivalent MAC" if dtype == "fp32" else "2500 TF/s dense 16-bit MFMA") +
            "; 8 TB/s HBM; frac = max(alg FLOPs / MFMA peak, alg bytes / HBM peak) / measured time",
            other_classes={k: v for k, v in per_class.items() if k != dom})
    rec["alg_hbm_gbs_whole_step"] = sum(v["bytes_per_launch"] * v["launches"] for v in rows.values()) / sampled / (step_ms * 1e-3) / 1e9
    return rec


def mixed_leg(args, dtype="bf16"):
    """Extra record: BASELINE configs[4]'s network (256x256x8 raster, default U-Net, 56,580,360 parameters) in mixed
    precision -- bf16 matrix-core products, 16-bit channel-blocked activations, fp32 statistics / accumulators --
    as denoising steps at --mixed-batch samples.  Priced against BOTH roofs: 2.5 PF/s dense bf16 MFMA and 8 TB/s HBM (the
    64 / 128-channel layers are HBM-bound in 16 bits)."""
    from drivescenegen_amd.configs import CFG5
    b = args.mixed_batch
    return forward_leg(args, CFG5, dtype, b, args.mixed_steps, args.ddim_steps,
                       "BASELINE configs[4] network: 256x256x8 map+agent raster, default U-Net (56,580,360 params), "
                       f"mixed {dtype}, DDIM step, batch {b} on 1 GPU", 353.58e9, "_bf16", MIXED_PMC_PATTERNS)


def cfg4_leg(args):
    """Extra record: BASELINE configs[3] at its own size -- 512x512x4 raster, the 6-level network with attention at 32^2 and
    16^2 (66,294,660 parameters), 100-step DDIM, batch 8 on one GPU, fp32-equivalent.  BASELINE calls it the 'HBM-bound
    conv path': four of its six levels are 64 / 128 channels wide at 512^2 ... 64^2."""
    from drivescenegen_amd.configs import CFG4
    return forward_leg(args, CFG4, "fp32", 8, 20, 100,
                       "BASELINE configs[3]: 512x512x4 high-res raster, U-Net with attention at 16^2 and 32^2 (6 levels, "
                       "66,294,660 params), 100-step DDIM (eta=0), batch 8 on 1 GPU, fp32-equivalent", CFG4_FLOPS_IMG, "_cfg4", CFG4_PMC_PATTERNS)


def train_ref_leg(args, steps=8):
    """Extra record: the reference's OWN training operating point (train.py:16,24,39-57: the 3-channel default network,
    mixed_precision='fp16', train_batch_size 14) through the Accelerator mirror's loop body (training_pipeline.py:70-91 =
    train_loop.train_step: add_noise, forward, MSE, scaled backward, unscale + clip 1.0, AdamW, cosine LR)."""
    import drivescenegen_amd as d
    from drivescenegen_amd import synth
    from drivescenegen_amd.configs import DEFAULT3, synth_weights
    from drivescenegen_amd.train_loop import train_steps
    dev = torch.device("cuda", _local_device())
    batch = 14
    acc = d.Accelerator(mixed_precision="fp16")
    net = synth_weights(d.UNet2DModel(**DEFAULT3)).to(dev)
    opt = d.AdamW(net.parameters(), lr=1e-4)
    lrs = d.get_cosine_schedule_with_warmup(optimizer=opt, num_warmup_steps=500, num_training_steps=50000)
    net, opt, lrs = acc.prepare(net, opt, lrs)
    sch = d.DDPMScheduler()
    x0 = torch.from_numpy(synth.synth_scene_rasters(batch, 3, 256, 256, 14555)).to(dev)
    # train_loop.train_steps = fit's inner loop: step k+1's host noise (training_pipeline.py:72) is drawn by a worker thread
    # while the GPU runs step k (same generator, same order, same values)
    # ONE pass of train_steps over warm-up + timed batches, as an epoch of `fit` is: the clock starts when the third step has been
    # queued (every later step's noise was drawn while its predecessor ran; only an epoch's first step draws on the critical path)
    clock, t0, warm = StepClock(dev), None, 3
    for i, loss in enumerate(train_steps(acc, net, sch, opt, lrs, [x0] * (warm + steps))):
        if i == warm - 1:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            clock.tick()
        elif i >= warm:
            clock.tick()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    assert torch.isfinite(loss).all()
    return {"metric": "training throughput, the reference's own configuration (train.py: default 3-channel U-Net, fp16 AMP + "
                      "GradScaler, batch 14)", "value": batch * steps / dt, "unit": "images/s", "ms_per_step": dt / steps * 1e3,
            "step_ms_spread": clock.spread(), "host_noise": "drawn one step ahead on a worker thread (train_loop.NoiseAhead)",
            "steps": steps, "batch": batch, "dtype": "f16 storage / MFMA, f32 accumulate and master weights",
            "loss": float(loss), "loss_scale": float(acc.scaler.get_scale()) if acc.scaler is not None else None,
            "peak_mem_gib": torch.cuda.max_memory_allocated(dev) / 2 ** 30}


def _write_synth_pngs(folder, count, size=512, seed=14555):
    """`count` synthetic 512x512 RGB scene rasters (flat background, lanes, agents + light sensor-like noise so that the
    files do not compress to nothing: ~400 KB each) as PNG files; returns the paths."""
    import numpy as np
    from PIL import Image
    from drivescenegen_amd import synth
    os.makedirs(folder, exist_ok=True)
    x = synth.synth_scene_rasters(count, 3, size, size, seed)
    img = ((x.transpose(0, 2, 3, 1) * 0.5 + 0.5) * 255).round().astype(np.uint8)
    noise = (synth.uniform01(seed, img.size, stream=5).reshape(img.shape) * 13).astype(np.int16) - 6
    img = np.clip(img.astype(np.int16) + noise, 0, 255).astype(np.uint8)
    paths = []
    for i in range(count):
        paths.append(os.path.join(folder, f"{i:04d}.png"))
        Image.fromarray(img[i]).save(paths[-1], compress_level=1)
    return paths


def train_e2e(net, opt, lrs, sch, cfg, batch, dtype, bare_ms, steps=6, warm=3):
    """The training loop END TO END at the config's own batch, the way the reference runs it
    (training_pipeline.py:70-97 over train.py:34-35's loader): PNG files on disk -> `GpuImageLoader` (native decode pool,
    pinned ring, H2D on a side stream, one resize + normalise kernel: dataset.py:32-50) -> `train_loop.train_steps`
    with (a) the reference's HOST noise draw (`torch.randn(batch.shape)`, training_pipeline.py:72, one step ahead on a worker
    thread: `NoiseAhead`) and (b) the opt-in library generator (`noise="device"`: dsg_add_noise_philox).  Reported next
    to the bare tape's step time: images/s of both modes, the loader's own rates, the host draw's cost."""
    import shutil
    import tempfile
    import drivescenegen_amd as d
    from drivescenegen_amd import imageops, train_loop
    dev = next(net.parameters()).device
    c = cfg["in_channels"]
    folder = tempfile.mkdtemp(prefix="dsg_e2e_")
    try:
        t0 = time.perf_counter()
        _write_synth_pngs(folder, 2 * batch)
        write_s = time.perf_counter() - t0
        print("[bench]   e2e: loader", file=sys.stderr, flush=True)
        loader = imageops.GpuImageLoader(os.path.join(folder, "*.png"), (256, 256), batch, shuffle=True, seed=7, device=dev)
        decode_rate = loader.decode_rate(3)
        pil_loader = imageops.GpuImageLoader(os.path.join(folder, "*.png"), (256, 256), batch, workers=1, native_png=False, device=dev)
        pil_rate = pil_loader.decode_rate(1)     # the reference's feeder: one thread, PIL

        def batches(n):
            """n batches [B, C, 256, 256] from epochs of the loader; the 3 decoded channels are tiled to the config's C"""
            k = 0
            while k < n:
                for x in loader:
                    yield x if c == 3 else x.repeat(1, (c + 2) // 3, 1, 1)[:, :c].contiguous()
                    k += 1
                    if k >= n:
                        return
        # the loader alone, GPU side included (decode + H2D + resize kernel), consumer does nothing else
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        n_img = sum(x.shape[0] for x in batches(6))
        torch.cuda.synchronize(dev)
        loader_rate = n_img / (time.perf_counter() - t0)
        # the host draw the reference makes per step (serial CPU generator)
        t0 = time.perf_counter()
        torch.randn((batch, c, 256, 256))
        draw_ms = (time.perf_counter() - t0) * 1e3

        acc = d.Accelerator(mixed_precision="no")   # (the net's compute dtype is already set; no GradScaler for fp32 / bf16)

        def run(noise):
            t_start, last, clock = None, None, StepClock(dev)
            for i, loss in enumerate(train_loop.train_steps(acc, net, sch, opt, lrs, batches(warm + steps), noise=noise)):
                if i == warm - 1:
                    torch.cuda.synchronize(dev)
                    t_start = time.perf_counter()
                if i >= warm - 1:
                    clock.tick()
                last = loss
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t_start
            assert torch.isfinite(last).all()
            return dt / steps * 1e3, float(last), clock.spread()
        print("[bench]   e2e: host noise", file=sys.stderr, flush=True)
        host_ms, host_loss, host_spread = run("host")
        print("[bench]   e2e: device noise", file=sys.stderr, flush=True)
        dev_ms, dev_loss, dev_spread = run(train_loop.DeviceNoise(seed=14555))
        return {"what": "PNG files -> GpuImageLoader -> train_steps, at the config's own batch; bare tape = device-resident x0 / noise / t",
                "batch": batch, "dtype": dtype, "bare_tape_ms": bare_ms, "bare_tape_images_s": batch / bare_ms * 1e3,
                "host_noise_ms": host_ms, "host_noise_images_s": batch / host_ms * 1e3,
                "device_noise_ms": dev_ms, "device_noise_images_s": batch / dev_ms * 1e3,
                "device_noise_vs_bare": bare_ms / dev_ms, "host_noise_vs_bare": bare_ms / host_ms,
                "host_draw_ms": draw_ms, "loader_images_s": loader_rate, "decode_pool_images_s": decode_rate,
                "decode_workers": loader.workers, "pil_one_thread_images_s": pil_rate,
                "png": "512x512 RGB, ~%d KB" % (os.path.getsize(os.path.join(folder, "0000.png")) // 1024),
                "host_noise_step_ms_spread": host_spread, "device_noise_step_ms_spread": dev_spread,
                "png_write_s": write_s, "loss_host": host_loss, "loss_device": dev_loss}
    finally:
        shutil.rmtree(folder, ignore_errors=True)


def train_leg(args, dtype="fp32", batch=64, steps=6):
    """Extra record: optimizer steps of the training loop (training_pipeline.py:70-91: add_noise, U-Net forward, MSE,
    backward, clip 1.0, AdamW, cosine LR) on BASELINE configs[2]'s network at `batch` samples on one GPU; images/s."""
    import drivescenegen_amd as d
    from drivescenegen_amd import synth
    from drivescenegen_amd.configs import CFG3, CFG5, synth_weights
    cfg = CFG3 if dtype == "fp32" else CFG5
    dev = torch.device("cuda", _local_device())
    net = synth_weights(d.UNet2DModel(**cfg)).to(dev).train().set_compute_dtype(dtype)
    opt = d.AdamW(net.parameters(), lr=1e-5)
    lrs = d.get_cosine_schedule_with_warmup(optimizer=opt, num_warmup_steps=500, num_training_steps=50000)
    sch = d.DDPMScheduler()
    c = cfg["in_channels"]
    x0 = torch.from_numpy(synth.synth_scene_rasters(batch, c, 256, 256, 14555)).to(dev)
    noise = torch.from_numpy(synth.normal(14556, (batch, c, 256, 256))).to(dev)
    t = torch.from_numpy((synth.uniform01(14557, batch) * 1000).astype("int64")).to(dev)

    def one():
        noisy = sch.add_noise(x0, noise, t)
        loss = d.mse_loss(net(noisy, t, return_dict=False)[0], noise)
        loss.backward()
        d.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        lrs.step()
        opt.zero_grad()
        return loss
    for _ in range(4):   # (at 60 GiB the caching allocator still grows during the second and, now and then, the third step: a
        one()            #  hipMalloc inside the timed region showed as ONE 216-ms step among 185-ms ones, profiles/r06_bench.json)
    torch.cuda.synchronize(dev)
    clock = StepClock(dev)
    t0 = time.perf_counter()
    clock.tick()
    for _ in range(steps):
        loss = one()
        clock.tick()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    assert torch.isfinite(loss.detach()).all()
    peak_gib = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    # one more step under the HIP-event profiler: the step's conv classes (forward + data-gradient convs share the forward
    # kernels' classes; the 3x3 weight gradient has its own), algorithmic FLOPs / bytes per launch as in the headline leg
    from drivescenegen_amd import _lib
    lib = _lib.load()
    lib.dsg_prof_enable(1)
    one()
    torch.cuda.synchronize(dev)
    base = 0 if dtype == "fp32" else 20
    rows = _prof_rows(lib, _lib, {base + 6: "conv3x3_fwd_and_dgrad", base + 10: "conv3x3_fwd_and_dgrad_two_wg_per_cu",
                                   base + 13: "conv3x3_plus_fused_shortcut",
                                   base + 7: "conv3x3_upsample", base + 8: "conv1x1", base + 2: "conv3x3_s2",
                                   base + 9: "conv3x3_wgrad", 5: "conv_wgrad_f32_mfma", 0: "conv3x3_s1_mfma_f32"})
    lib.dsg_prof_enable(0)
    e2e = None
    if not getattr(args, "no_e2e", False):
        try:
            e2e = train_e2e(net, opt, lrs, sch, cfg, batch, dtype, dt / steps * 1e3)
        except Exception as e:  # noqa: BLE001  (reported, never hidden)
            e2e = {"error": f"{type(e).__name__}: {e}"}
    del net, opt
    torch.cuda.empty_cache()
    rec = {"metric": "training images/sec (fwd + bwd + clip + AdamW) -- BARE TAPE: device-resident x0 / noise / t, no loader, no "
                     "noise draw; the loop end to end is `e2e`", "value": batch * steps / dt, "unit": "images/s",
           "ms_per_step": dt / steps * 1e3, "step_ms_spread": clock.spread(), "steps": steps, "dtype": dtype,
           "peak_mem_gib": peak_gib,
           "config": {"workload": f"BASELINE configs[{2 if dtype == 'fp32' else 4}]: 256x256x{c} raster, DDPM training step (add_noise, "
                                  f"fwd, MSE, bwd, clip 1.0, AdamW, cosine LR), batch {batch} per GPU (the config's own), {dtype}, "
                                  "one GPU's share of the data-parallel step (no all-reduce at N = 1); bare tape, "
                                  "device-resident inputs", "batch": batch},
           "e2e": e2e, "kernels": rows}
    # roofline of the dominant BACKWARD kernel (the 3x3 weight gradient) and of the conv class that carries forward and
    # data gradients, priced like the headline: fp32-equivalent = 2500 / 3 TF/s-eq, 16-bit = 2500 TF/s; HBM 8 TB/s
    peak = PEAK_F16_TFLOPS / (3.0 if dtype == "fp32" else 1.0)
    step_ms = dt / steps * 1e3

    def roof(row, kernel, pmc_pat):
        t_mfma = row["flops_per_launch"] / (peak * 1e12)
        t_hbm = row["bytes_per_launch"] / (PEAK_HBM_GBS * 1e9)
        bound = "hbm" if t_hbm > t_mfma else "mfma"
        return dict(bound=bound, kernel=kernel, achieved=row["alg_gbs"] if bound == "hbm" else row["tflops"],
                    peak=PEAK_HBM_GBS if bound == "hbm" else peak, unit="GB/s" if bound == "hbm" else "TFLOP/s",
                    frac=max(t_mfma, t_hbm) / (row["avg_ms"] * 1e-3), mfma_tflops=row["tflops"], alg_gbs=row["alg_gbs"],
                    avg_launch_ms=row["avg_ms"], launches=row["launches"], time_share=row["total_ms"] / step_ms,
                    **pmc_class_traffic(pmc_pat, "_train_" + dtype))
    wg, fw = rows.get("conv3x3_wgrad"), rows.get("conv3x3_fwd_and_dgrad")
    if wg:
        rec["roofline"] = roof(wg, "dsg::conv_wgrad_h2w_kernel (cout % 128 == 0) + dsg::conv_wgrad_h2_kernel (64-cout layers)"
                               if dtype == "fp32" else "dsg::conv_wgrad16_kernel<*, 3, *>",
                               r"conv_wgrad_h2w?_kernel" if dtype == "fp32" else r"conv_wgrad16_kernel<\d, 3")
        if fw:
            rec["roofline"]["second_kernel"] = roof(fw, "dsg::conv_h2_kernel<0, *, 3, *> (forward and data-gradient 3x3 convs)",
                                                    r"conv_h2_kernel<0, [24], 3, [02], 4, [12], [03], (64|128), " +
                                                    ("0" if dtype == "fp32" else "[12]"))
    return rec


def train_ddp_leg(args, rank, world, dtype, batch, steps):
    """Record for N > 1 (every rank calls it): the data-parallel training step north_star names -- BASELINE configs[2]
    (fp32, 64 per GPU) / configs[4] (bf16, 128 per GPU) through `Accelerator` + `GradBuckets` (train.py:121-122,
    training_pipeline.py:59-61,86): rank-0 broadcast, ~25-MB reverse-order buckets all-reduced (AVG) on RCCL from INSIDE the
    backward walk.  Three timings of the same step on every rank (max over ranks): `local` (buckets switched off: the bare
    tape), `overlap` (the product path), `deferred` (DSG_DDP_OVERLAP=0: the same buckets launched after the walk); exposed
    communication = mode - local.  Plus the self-check: the all-reduced gradient slab of one backward in both modes, bitwise."""
    import torch.distributed as dist
    import drivescenegen_amd as d
    from drivescenegen_amd import synth
    from drivescenegen_amd.autograd import get_train_state
    from drivescenegen_amd.configs import CFG3, CFG5, synth_weights
    cfg = CFG3 if dtype == "fp32" else CFG5
    acc = d.Accelerator(mixed_precision="no" if dtype == "fp32" else dtype)
    dev = acc.device
    net = synth_weights(d.UNet2DModel(**cfg)).train()
    opt = d.AdamW(net.parameters(), lr=1e-5)
    lrs = d.get_cosine_schedule_with_warmup(optimizer=opt, num_warmup_steps=500, num_training_steps=50000)
    net, opt, lrs = acc.prepare(net, opt, lrs)
    sch = d.DDPMScheduler()
    c = cfg["in_channels"]
    # rank r's shard of the global batch: its own samples, noise and timesteps
    x0 = torch.from_numpy(synth.synth_scene_rasters(batch, c, 256, 256, 14555 + rank)).to(dev)
    noise = torch.from_numpy(synth.normal(24556 + rank, (batch, c, 256, 256))).to(dev)
    t = torch.from_numpy((synth.uniform01(34557 + rank, batch) * 1000).astype("int64")).to(dev)

    def fwd_bwd():
        with acc.accumulate(net):
            loss = d.mse_loss(net(sch.add_noise(x0, noise, t), t, return_dict=False)[0], noise)
            acc.backward(loss)
        return loss

    def one():
        with acc.accumulate(net):
            loss = d.mse_loss(net(sch.add_noise(x0, noise, t), t, return_dict=False)[0], noise)
            acc.backward(loss)
            acc.clip_grad_norm_(net.parameters(), 1.0)
            opt.step()
            lrs.step()
            opt.zero_grad()
        return loss

    acc.ddp_overlap = True
    for _ in range(2):
        one()           # (creates the train state, the gradient slab and the buckets; the allocator settles)
    buckets = acc._buckets
    assert buckets is not None and buckets.active, "train_ddp_leg: no active gradient buckets (process group missing?)"

    def timed(mode):
        buckets.active = mode != "local"
        acc.ddp_overlap = mode != "deferred"
        one()
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = one()
        torch.cuda.synchronize(dev)
        own = (time.perf_counter() - t0) / steps * 1e3
        assert torch.isfinite(loss.detach()).all()
        box = torch.tensor([own], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(box) for _ in range(world)]
        dist.all_gather(gathered, box)
        per = [float(g) for g in gathered]
        buckets.active, acc.ddp_overlap = True, True
        return max(per), per
    ms, per_rank = {}, {}
    for mode in ("local", "overlap", "deferred"):
        ms[mode], per_rank[mode] = timed(mode)
    # self-check: one backward from the same weights, gradients all-reduced from inside the walk vs after it -- bitwise
    st = get_train_state(net)
    slabs = {}
    for overlap in (True, False):
        acc.ddp_overlap = overlap
        st.grad_flat.zero_()
        fwd_bwd()
        torch.cuda.synchronize(dev)
        slabs[overlap] = st.grad_flat.clone()
        order = list(buckets.last_launch_order)
    acc.ddp_overlap = True
    st.grad_flat.zero_()
    same = torch.tensor([1.0 if torch.equal(slabs[True], slabs[False]) else 0.0], device=dev)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    peak_gib = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    rec = {"metric": "data-parallel training images/sec (global), gradient all-reduce overlapped with backward",
           "value": batch * world / ms["overlap"] * 1e3, "unit": "images/s", "batch_per_gpu": batch, "dtype": dtype,
           "ms_per_step": ms["overlap"], "ms": ms, "per_rank_ms": per_rank,
           "exposed_comm_ms": {"overlap": ms["overlap"] - ms["local"], "deferred": ms["deferred"] - ms["local"]},
           "buckets": len(buckets.buckets), "bucket_mb": 25, "grad_slab_mb": st.grad_flat.numel() * 4 / 2 ** 20,
           "launch_order": order, "world": world, "backend": dist.get_backend(), "avg_native": buckets.avg_native,
           "selfcheck_bitwise": bool(float(same) == 1.0), "steps": steps, "peak_mem_gib": peak_gib,
           "config": {"workload": f"BASELINE configs[{2 if dtype == 'fp32' else 4}]: 256x256x{c}, DDPM training step, batch {batch} per "
                                  f"GPU x {world} ranks, {dtype}, grads all-reduced (AVG) in ~25-MB buckets from inside the backward walk"}}
    del net, opt, slabs
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def small_batch_leg(args):
    """Extra record: the reference's own sampling calls -- DDPM ancestral steps at batch 1 (training_pipeline.py:26-32)
    and batch 5 (generation.py:12-20) of the 3-channel default network; ms per denoising step."""
    import drivescenegen_amd as d
    from drivescenegen_amd import synth
    from drivescenegen_amd.configs import DEFAULT3, synth_weights
    dev = torch.device("cuda", _local_device())
    net = synth_weights(d.UNet2DModel(**DEFAULT3)).to(dev).eval().requires_grad_(False)
    sch = d.DDPMScheduler()
    sch.set_timesteps(750)
    out = {}
    for b in (1, 5):
        x = torch.from_numpy(synth.normal(14555, (b, 3, 256, 256), stream=3)).to(dev)
        nz = torch.from_numpy(synth.normal(14555, (b, 3, 256, 256), stream=4)).to(dev)
        ts = [int(t) for t in sch.timesteps[:55]]

        def step(t, x):
            return sch.step(net(x, t).sample, t, x, variance_noise=nz).prev_sample
        for t in ts[:5]:
            x = step(t, x)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for t in ts[5:]:
            x = step(t, x)
        torch.cuda.synchronize(dev)
        out[f"batch{b}_ms_per_step"] = (time.perf_counter() - t0) / len(ts[5:]) * 1e3
    # the reference's calls END TO END (VERDICT r04 missing item 3): everything a caller waits for -- x_T, 750 forwards and
    # scheduler steps, each step's noise (device RNG with no generator; the seeded CPU generator's draws + PCIe for evaluate),
    # the per-step host work, post-processing, the D2H copy and numpy_to_pil -- next to 750 x the per-step time above
    pipe = d.DDPMPipeline(unet=net, scheduler=d.DDPMScheduler())
    pipe(batch_size=1, num_inference_steps=20)     # (warm: plan, workspaces, pinned buffers)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    images = pipe(batch_size=5, num_inference_steps=750).images                       # generation.py:14-20
    t_gen = time.perf_counter() - t0
    assert len(images) == 5
    t0 = time.perf_counter()
    arr = pipe(num_inference_steps=750, batch_size=1, generator=torch.manual_seed(14555), output_type="np.array",
               return_dict=False)[0]                                                  # training_pipeline.py:26-32
    t_eval = time.perf_counter() - t0
    assert arr.shape == (1, 256, 256, 3)
    out["whole_call"] = {
        "generation_py_batch5_750_steps_s": t_gen, "generation_py_ms_per_step": t_gen / 750 * 1e3,
        "generation_py_over_750_x_step": t_gen * 1e3 / (750 * out["batch5_ms_per_step"]),
        "evaluate_batch1_750_steps_cpu_generator_s": t_eval, "evaluate_ms_per_step": t_eval / 750 * 1e3,
        "evaluate_over_750_x_step": t_eval * 1e3 / (750 * out["batch1_ms_per_step"]),
        "note": "wall time of DDPMPipeline.__call__ as the reference calls it, PIL / numpy output included"}
    out["config"] = {"workload": "DriveSceneGen default U-Net (train.py:39-57, 3 channels), 750-step DDPM, batch 1 "
                                 "(evaluate) and batch 5 (generation.py), fp32-equivalent"}
    return out


def _pmc_stamp(d):
    """Where a quoted counter file came from: the git head its collection ran on (tools/gpu.sh writes it before the call)
    and the profiled command -- a kernel whose template arguments changed since then no longer matches and reads None;
    one whose code changed under the same name shows here as an older head."""
    return dict(traffic_git_head=d.get("git_head"), traffic_cmd=(d.get("source") or "").split("over: ")[-1] or None)


def pmc_traffic(kernel, suffix=""):
    """HBM bytes per launch of `kernel` from the committed PMC pass (tools/pmc_bench.sh -> profiles/*_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same bench command line, FETCH_SIZE doubled as
    the gfx950 guide prescribes).  bench.py cannot collect counters on itself; the newest committed file is quoted,
    with its name, next to the live HIP-event numbers.  None when no such file travels with the repo."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"*_pmc_traffic{suffix}.json")))
    if not files:
        return dict(traffic=None)
    try:
        d = json.load(open(files[-1]))
        # (the template argument list grew over the rounds: match the instantiation by its leading arguments)
        k = next(v for name, v in sorted(d["kernels"].items()) if name.startswith(kernel.rstrip(">")))
        return dict(traffic=k["hbm_bytes_per_launch"], traffic_unit="bytes/launch (HBM read + write, mean over launches)",
                    traffic_read=k["fetch_bytes_per_launch"], traffic_write=k["write_bytes_per_launch"],
                    traffic_source="profiles/" + os.path.basename(files[-1]), **_pmc_stamp(d))
    except (KeyError, ValueError, OSError, StopIteration):
        return dict(traffic=None)


def pmc_class_traffic(pattern, suffix):
    """Launch-weighted mean HBM bytes per launch over every kernel of the newest profiles/*_pmc_traffic<suffix>.json
    whose name matches `pattern` (a kernel class served by several instantiations)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"*_pmc_traffic{suffix}.json")))
    if not files:
        return dict(traffic=None)
    try:
        d = json.load(open(files[-1]))
        ks = [v for name, v in d["kernels"].items() if re.search(pattern, name)]
        n = sum(k["launches"] for k in ks)
        if not n:
            return dict(traffic=None)
        return dict(traffic=sum(k["hbm_bytes_per_launch"] * k["launches"] for k in ks) / n,
                    traffic_unit="bytes/launch (HBM read + write, launch-weighted mean over the class's instantiations)",
                    traffic_source="profiles/" + os.path.basename(files[-1]), **_pmc_stamp(d))
    except (KeyError, ValueError, OSError):
        return dict(traffic=None)


def pmc_mfma(kernel, suffix=""):
    """Matrix-pipe utilisation of `kernel` from the committed SQ-counter pass (tools/pmc_mfma.sh ->
    profiles/*_pmc_mfma.json): MFMA instructions issued x 32 cycles / (GPU cycles x 1024 SIMDs), independent of the
    clock the chip throttles to.  Quoted next to the live figures like the HBM traffic; {} when no file travels."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"*_pmc_mfma{suffix}.json")))
    if not files:
        return {}
    try:
        d = json.load(open(files[-1]))
        k = next(v for name, v in sorted(d["kernels"].items()) if name.startswith(kernel.rstrip(">")))
        return dict(mfma_pipe_util=k["mfma_util_issued"], mfma_pipe_util_source="profiles/" + os.path.basename(files[-1]),
                    mfma_pipe_util_git_head=d.get("git_head"))
    except (KeyError, ValueError, OSError, StopIteration):
        return {}


def summary_of(out):
    """Flat digest of the line, printed as its last key: value / ms_per_step / roofline.frac of the headline and of every
    extra record, and the CPU baseline."""
    def short(r):
        if not isinstance(r, dict):
            return r
        if "error" in r:
            return {"error": r["error"][:200]}
        d = {k: r[k] for k in ("value", "unit", "ms_per_step", "batch1_ms_per_step", "batch5_ms_per_step", "peak_mem_gib") if k in r}
        if isinstance(r.get("whole_call"), dict):
            d.update({k: round(v, 4) for k, v in r["whole_call"].items() if k.endswith("_ms_per_step") or k.endswith("_x_step")})
        if isinstance(r.get("config"), dict) and "batch" in r["config"]:
            d["batch"] = r["config"]["batch"]
        rf = r.get("roofline")
        if isinstance(rf, dict):
            d["roofline"] = {k: rf.get(k) for k in ("bound", "frac", "achieved", "unit", "traffic", "time_share") if k in rf}
            for sub in ("second_kernel", "fused_shortcut_kernel", "fused_shortcut_two_wg_kernel"):
                if isinstance(rf.get(sub), dict):
                    d["roofline"][sub + "_frac"] = rf[sub].get("frac")
        if isinstance(r.get("step_ms_spread"), dict):
            d["step_ms_min_med_max"] = [round(r["step_ms_spread"][k], 3) for k in ("min", "median", "max")]
        return d
    s = {"headline": short(out)}
    for name, rec in (out.get("extra_records") or {}).items():
        s[name] = short(rec)
    # the training loop END TO END (files -> loader -> train_steps) next to the bare tape, one entry per training config
    e2e = {}
    for name, rec in (out.get("extra_records") or {}).items():
        e = rec.get("e2e") if isinstance(rec, dict) else None
        if isinstance(e, dict):
            key = f"{e.get('dtype', name)}_b{e.get('batch', '')}" if "error" not in e else name
            e2e[key] = ({"error": e["error"][:160]} if "error" in e else
                        {k: round(e[k], 3) for k in ("bare_tape_images_s", "device_noise_images_s", "host_noise_images_s",
                                                     "device_noise_vs_bare", "host_noise_vs_bare", "host_draw_ms", "loader_images_s",
                                                     "decode_pool_images_s", "pil_one_thread_images_s") if k in e})
    if e2e:
        s["train_e2e"] = e2e
    for name, rec in (out.get("train_ddp") or {}).items():
        s["train_ddp_" + name] = ({"error": rec["error"][:200]} if "error" in rec else
                                  {k: rec[k] for k in ("value", "unit", "batch_per_gpu", "world", "backend", "ms", "exposed_comm_ms",
                                                       "buckets", "selfcheck_bitwise") if k in rec})
    if "cpu_baseline" in out:
        s["cpu_baseline"] = {k: out["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind")}
    return s


COMPACT_LIMIT = 6000   # bytes; the driver's parser took r03's 15 KB and not r04's 25.6 KB -- stay far below both
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_git_head",
                 "mfma_pipe_util", "avg_launch_ms", "launches", "alg_flops_per_launch", "alg_gbs", "time_share")


def write_full_record(out, path=None):
    """The whole record as JSON under gpurun_out/ (merged back from the GPU box); returns the path or None."""
    path = os.path.abspath(path) if path else os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f)
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def compact_line(out, full_path=None):
    """The ONE stdout line: the contract's top-level keys, `roofline` reduced to its scalars, `cpu_baseline`, and `summary`
    (one short object per extra record).  Always below COMPACT_LIMIT bytes: text fields are cut, never the numbers."""
    def cut(v, n):
        return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "~"
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")
    line = {k: out.get(k) for k in keys}
    line["dtype"] = cut(line["dtype"], 160)
    if isinstance(line["config"], dict):
        line["config"] = {k: cut(v, 200) for k, v in line["config"].items()}
    rf = out.get("roofline")
    line["roofline"] = {k: rf.get(k) for k in ROOFLINE_KEYS if k in rf} if isinstance(rf, dict) else None
    if isinstance(rf, dict):
        for sub in ("second_kernel", "fused_shortcut_kernel", "fused_shortcut_two_wg_kernel"):
            if isinstance(rf.get(sub), dict):
                line["roofline"][sub] = {k: rf[sub].get(k) for k in ("frac", "avg_launch_ms", "launches", "time_share", "traffic")}
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = dict(cb, sample=cut(cb.get("sample"), 200)) if isinstance(cb, dict) else None
    for k in ("step_ms_spread", "rccl_world", "dist_backend", "per_rank_ms_per_step", "whole_net_tflops"):
        if k in out:
            line[k] = out[k]
    line["full_record"] = full_path
    line["summary"] = {k: v for k, v in (out.get("summary") or {}).items() if k not in ("headline", "cpu_baseline")}
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= COMPACT_LIMIT:     # (only a pathological record gets here: drop the digest's optional fields, then the digest)
        line["summary"] = {k: ({kk: vv for kk, vv in v.items() if kk in ("value", "unit", "ms_per_step", "error")}
                               if isinstance(v, dict) else v) for k, v in line["summary"].items()}
        text = json.dumps(line, separators=(",", ":"))
    if len(text) >= COMPACT_LIMIT:
        line["summary"] = None
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < COMPACT_LIMIT, len(text)
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="samples per GPU (BASELINE configs[1]: 16)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--no-prof", action="store_true", help="disable the HIP-event roofline leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--prof-dump", default=None, help="write the per-launch HIP-event records (CSV) here")
    ap.add_argument("--pure-f32", action="store_true",
                    help="disable the fp16x2-split conv path: every contraction on the f32 MFMA (A/B reference)")
    ap.add_argument("--separate-gn-stats", action="store_true",
                    help="GroupNorm statistics by a pass of their own instead of the producing conv's epilogue (A/B)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads for the CPU oracle (16 is the fastest setting on the 2x64-core GPU box: "
                         "32/64/128/256 threads run 1.1x/2x/4.4x/36x slower)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra records (mixed-precision forward, training step, small-batch sampling)")
    ap.add_argument("--no-ddp", action="store_true", help="skip the data-parallel training record that runs whenever a process group exists")
    ap.add_argument("--ddp-fp32-batch", type=int, default=64, help="train_ddp: BASELINE configs[2], 64 per GPU (0: skip)")
    ap.add_argument("--ddp-bf16-batch", type=int, default=128, help="train_ddp: BASELINE configs[4], 128 per GPU (0: skip)")
    ap.add_argument("--ddp-steps", type=int, default=3)
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (files -> loader -> train_steps) part of the training records")
    ap.add_argument("--train-fp32-batch", type=int, default=64, help="BASELINE configs[2]: 64 per GPU")
    ap.add_argument("--train-bf16-batch", type=int, default=128, help="BASELINE configs[4]: 128 per GPU")
    ap.add_argument("--mixed-batch", type=int, default=64)
    ap.add_argument("--mixed-steps", type=int, default=20)
    ap.add_argument("--full-record", default=None,
                    help="where the FULL record (every class row of every leg) is written; default gpurun_out/bench_full.json")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # asked for N GPUs without a launcher: start one rank per GPU ourselves (the same command line the driver uses)
        import socket
        import subprocess
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible")
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # (DSG_FORCE_COLLECTIVES=1 under a launcher: the RCCL group, barrier and max-over-ranks also run at WORLD_SIZE 1)
    if world > 1 or (os.environ.get("DSG_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(_local_device())
        # RCCL ("nccl" on ROCm).  DSG_DIST_BACKEND=gloo is the test hook that lets two ranks share one GPU (tests/test_gpu_rccl_one_rank.py)
        torch.distributed.init_process_group(os.environ.get("DSG_DIST_BACKEND", "nccl"))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    dt, prof, per_rank, spread = gpu_leg(args, rank, world)
    train_ddp = None
    if torch.distributed.is_initialized() and not args.no_ddp:
        # N > 1 (or the forced one-rank RCCL group): the data-parallel TRAINING step on every rank -- what north_star names
        # ("RCCL all-reduce over xGMI overlapped with backward"); the inference leg above has no collective to measure
        train_ddp = {}
        for name, dtype, batch in (("fp32", "fp32", args.ddp_fp32_batch), ("bf16", "bf16", args.ddp_bf16_batch)):
            if batch <= 0:
                continue
            try:
                train_ddp[name] = train_ddp_leg(args, rank, world, dtype, batch, args.ddp_steps)
            except Exception as e:  # noqa: BLE001  (reported; the other ranks fail the same way or the barrier below hangs loudly)
                train_ddp[name] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        n_img_steps = args.batch * world * args.steps
        value = n_img_steps / dt
        roofline = None
        if "conv3x3_s1_mfma_f16x2split" in prof:
            # dominant kernel: 3x3 conv with fp32-equivalent accuracy on the f16 matrix cores, 3 MFMA products per
            # fp32-equivalent MAC -> the ceiling for ALGORITHMIC (fp32-equivalent) FLOPs is the dense f16 peak / 3
            dom = prof["conv3x3_s1_mfma_f16x2split"]
            peak = PEAK_F16_TFLOPS / 3.0
            roofline = dict(bound="mfma", kernel="dsg::conv_h2_kernel<0, 4, 3, 2, 4, 1, 3, 64, 0, 0, 0, 0>", achieved=dom["tflops"], peak=peak,
                            unit="TFLOP/s", frac=dom["tflops"] / peak, **pmc_traffic("dsg::conv_h2_kernel<0, 4, 3, 2, 4, 1, 3, 64, 0, 0, 0, 0>"),
                            **pmc_mfma("dsg::conv_h2_kernel<0, 4, 3, 2, 4, 1, 3, 64, 0, 0, 0, 0>"),
                            peak_note="2500 TF/s dense f16 MFMA / 3 products per fp32-equivalent MAC (fp16x2 split); "
                                      "issued MFMA rate = 3 x achieved; the pure-fp32 MFMA peak is 157.3",
                            issued_mfma_tflops=3.0 * dom["tflops"],
                            frac_of_f32_mfma_peak=dom["tflops"] / PEAK_F32_TFLOPS,
                            avg_launch_ms=dom["avg_ms"], launches=dom["launches"],
                            alg_flops_per_launch=dom["flops_per_launch"],
                            alg_gbs=dom["alg_gbs"], sampled_every_nth_step=PROF_EVERY,
                            time_share=dom["total_ms"] * 1e-3 * args.steps / len(range(0, args.steps, PROF_EVERY)) / dt)
            ws = prof.get("conv3x3_s1_mfma_f16x2split_two_wg_per_cu")
            if ws:  # the same convs at the 64- / 128-channel levels (cin <= 128): 8-row tiles, two workgroups per CU
                roofline["second_kernel"] = dict(
                    kernel="dsg::conv_h2_kernel<0, 2, 3, 2, 4, 2, 3, 64, 0, 1, 0, 0>", achieved=ws["tflops"], frac=ws["tflops"] / peak,
                    avg_launch_ms=ws["avg_ms"], launches=ws["launches"], alg_flops_per_launch=ws["flops_per_launch"],
                    alg_gbs=ws["alg_gbs"],
                    time_share=ws["total_ms"] * 1e-3 * args.steps / len(range(0, args.steps, PROF_EVERY)) / dt,
                    **pmc_traffic("dsg::conv_h2_kernel<0, 2, 3, 2, 4, 2, 3, 64, 0, 1, 0, 0>"))
            # the resnets' conv2 with the 1x1 shortcut contracted in the same kernel (FLOPs / bytes of both convs)
            for key, cls, kern in (("fused_shortcut_kernel", "conv3x3_plus_fused_shortcut_f16x2split", "dsg::conv_h2_kernel<0, 4, 3, 2, 4, 1, 3, 64, 0, 0, 1, 0>"),
                                   ("fused_shortcut_two_wg_kernel", "conv3x3_plus_fused_shortcut_f16x2split_two_wg_per_cu",
                                    "dsg::conv_h2_kernel<0, 2, 3, 2, 4, 2, 3, 64, 0, 1, 1, 0>")):
                row = prof.get(cls)
                if row:
                    roofline[key] = dict(
                        kernel=kern, achieved=row["tflops"], frac=row["tflops"] / peak, avg_launch_ms=row["avg_ms"],
                        launches=row["launches"], alg_flops_per_launch=row["flops_per_launch"], alg_gbs=row["alg_gbs"],
                        hbm_frac=row["alg_gbs"] / PEAK_HBM_GBS,
                        time_share=row["total_ms"] * 1e-3 * args.steps / len(range(0, args.steps, PROF_EVERY)) / dt,
                        **pmc_traffic(kern))
        elif "conv3x3_s1_mfma_f32" in prof:
            dom = prof["conv3x3_s1_mfma_f32"]
            roofline = dict(bound="mfma", kernel="dsg::conv_mfma_kernel<3,1,0,2,*>", achieved=dom["tflops"],
                            peak=PEAK_F32_TFLOPS, unit="TFLOP/s", frac=dom["tflops"] / PEAK_F32_TFLOPS, traffic=None,
                            avg_launch_ms=dom["avg_ms"], launches=dom["launches"],
                            alg_flops_per_launch=dom["flops_per_launch"],
                            time_share=dom["total_ms"] * 1e-3 / dt)
        flops_img = 352.98e9  # SURVEY 8d, cfg2 forward
        out = {
            "metric": "denoising-steps/sec (U-Net fwd) on 256x256 BEV rasters", "value": value,
            "unit": "image-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "step_ms_spread": spread, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (3x3 and 1x1 convs: fp32-equivalent contraction as an fp16x2 split on the f16 MFMA, fp32 "
                     "accumulate; everything else f32)" if "conv3x3_s1_mfma_f16x2split" in prof else "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 256x256x4 BEV raster, DriveSceneGen default U-Net "
                                   "(56,575,748 params), 50-step DDIM (eta=0), batch 16 per GPU, fp32",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "ddim_steps": args.ddim_steps, "parallelism": f"sample-sharded x{world}, no collective"},
            # the process group the timing barrier / max ran on (RCCL = torch's "nccl" backend on ROCm) and each rank's own
            # wall time per step: `ms_per_step` is their maximum
            "rccl_world": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 0,
            "dist_backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
            "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
            "whole_net_tflops": value * flops_img / 1e12,
            "whole_net_frac_of_f32_peak": value * flops_img / 1e12 / (PEAK_F32_TFLOPS * world),
            "roofline": roofline,
            "kernels": prof,
            "class_notes": {k: v for k, v in CLASS_NOTES.items() if k in prof},
        }
        if train_ddp is not None:
            out["train_ddp"] = train_ddp
        if not args.no_extras and world == 1:  # bounded extra legs, N = 1 only; a failure is reported, never hidden
            extras = {}
            for name, fn in (("mixed_bf16", lambda: mixed_leg(args, "bf16")), ("configs3_512", lambda: cfg4_leg(args)),
                             ("train_fp32", lambda: train_leg(args, "fp32", batch=args.train_fp32_batch)),
                             ("train_bf16", lambda: train_leg(args, "bf16", batch=args.train_bf16_batch)),
                             ("train_fp16_reference_point", lambda: train_ref_leg(args)),
                             ("small_batch_sampling", lambda: small_batch_leg(args))):
                print(f"[bench] extra record: {name}", file=sys.stderr, flush=True)   # (a GPU fault kills the process: say where)
                try:
                    extras[name] = fn()
                except Exception as e:  # noqa: BLE001
                    extras[name] = {"error": f"{type(e).__name__}: {e}"}
                import gc
                gc.collect()
                torch.cuda.empty_cache()
            out["extra_records"] = extras
        if not args.no_cpu and world == 1:  # (rank 0 at N = 1 only: the other ranks of a multi-GPU run would wait for it)
            out["cpu_baseline"] = cpu_leg(args)
        out["summary"] = summary_of(out)
        # the FULL record (every class row of every leg, ~25 KB) goes to a file and to stderr; stdout's last line is the
        # compact line the driver parses (round 4's 25.6-KB line was not parsed: BENCH_r04.json "parsed": null)
        full_path = write_full_record(out, args.full_record)
        sys.stderr.write(json.dumps(out) + "\n")
        sys.stderr.flush()
        print(compact_line(out, full_path), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
