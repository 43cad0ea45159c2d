"""The engine next to OTHER PROCESSES on the same GPU: results must not depend on who else is running.

Found in round 4 by running two ranks on the one GPU (tests/test_gpu_rccl_one_rank.py): the fp32 training forward was bit-stable
alone and changed values in ~90 % of its runs as soon as another process of the library ran on the GPU.  One kernel was behind it --
conv_fewout_kernel (conv_out of the [N,C,H,W] tape) -- and behind that one instruction form: on gfx950 a packed fp32 VALU op whose
op_sel takes a HIGH dword for its low lane (v_pk_fma_f32 ... op_sel:[0,1,0]; hipcc emits it by itself) returns wrong values on lanes
48-63 while another wave on the same SIMD issues the 16-deep v_mfma_f32_32x32x16_f16 (csrc/conv.hip, DESIGN section 10,
profiles/r04_race_under_load.txt, tools/probes/).  tests/test_isa_policy.py keeps the form out of the built library; a kernel with such
a latent hazard passes every test that runs alone, so this file runs the engine's legs under a steady background load: two other
processes looping U-Net forwards (their pointwise convs are the aggressor kind).
  * single ops and whole forwards, repeated: every repetition must give the same bits;
  * training steps (forward, backward, clip, AdamW): the loaded run must be the solo run, bit for bit.
Reference: the loops of training_pipeline.py:59-101 and generation.py:14-20 share the GPU with whatever else a node runs."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import configs, ops, synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * scale).astype(np.float32)).to(DEV)


def _train_run(cfg, batch, dtype, steps):
    """loss and checksums of gradients / parameters per step of a short training run (deterministic inputs)"""
    ss = cfg["sample_size"]
    h, w = (ss, ss) if isinstance(ss, int) else ss
    c = cfg["in_channels"]
    net = configs.synth_weights(d.UNet2DModel(**cfg)).to(DEV).train().set_compute_dtype(dtype)
    opt = d.AdamW(net.parameters(), lr=1e-4)
    sch = d.DDPMScheduler()
    x0 = torch.from_numpy(synth.synth_scene_rasters(batch, c, h, w, 1)).to(DEV)
    noise = torch.from_numpy(synth.normal(2, (batch, c, h, w))).to(DEV)
    t = torch.tensor([3, 250, 600, 999] * ((batch + 3) // 4), device=DEV)[:batch]
    out = []
    for _ in range(steps):
        loss = d.mse_loss(net(sch.add_noise(x0, noise, t), t, return_dict=False)[0], noise)
        loss.backward()
        g = float(sum(p.grad.double().abs().sum() for p in net.parameters() if p.grad is not None))
        d.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        out.append((float(loss.detach()).hex(), g.hex(), float(sum(p.detach().double().abs().sum() for p in net.parameters())).hex()))
    return out


def _forward_sums(cfg, batch, dtype, reps, train):
    ss = cfg["sample_size"]
    h, w = (ss, ss) if isinstance(ss, int) else ss
    net = configs.synth_weights(d.UNet2DModel(**cfg)).to(DEV)
    net = (net.train() if train else net.eval().requires_grad_(False)).set_compute_dtype(dtype)
    x = torch.from_numpy(synth.normal(3, (batch, cfg["in_channels"], h, w))).to(DEV)
    t = torch.tensor([980, 20, 500, 3][:batch], device=DEV)
    return [float(net(x, t, return_dict=False)[0].detach().double().abs().sum()).hex() for _ in range(reps)]


def _conv_out_runs(reps):
    """conv_out of the [N,C,H,W] tape (cout <= 4: conv_fewout_kernel) against the VALU reference kernel"""
    bsz, c, cout, h, w = 4, 32, 3, 64, 64
    x0, wt, bias = _t(1, (bsz, c, h, w)), _t(3, (cout, c, 3, 3), 0.06), _t(4, (cout,), 0.1)
    wf = ops.relayout_conv_weight(wt)
    ssf, _ = ops.gn_scale_shift_train(x0, 1 + _t(5, (c,), 0.1), _t(6, (c,), 0.1), 32, 1e-5)
    kw = dict(ksize=3, gn_scale_shift=ssf, silu=True, cout=cout)
    ref = ops.conv2d_fused(x0, wf, bias, direct=True, **kw)
    worst, first = 0.0, None
    same = True
    for _ in range(reps):
        y = ops.conv2d_fused(x0, wf, bias, **kw)
        worst = max(worst, float((y - ref).abs().max()))
        first = y.clone() if first is None else first
        same = same and torch.equal(y, first)
    return worst, same


def _streaming_rows(reps):
    """rows f1 / f4 (the two other kernels hipcc had given the vulnerable packed form): repeated runs must return one result"""
    from drivescenegen_amd import imageops
    from drivescenegen_amd import rasterization as rz
    rng = np.random.default_rng(7)
    imgs = torch.from_numpy(rng.integers(0, 256, (4, 96, 80, 3), dtype=np.uint8)).to(DEV)
    fimg = torch.from_numpy(rng.random((4, 96, 80, 3), dtype=np.float32)).to(DEV)
    n = 400   # boxes [N][9] = (cx, cy, ux, uy, hx, hy, r, g, b) in draw order (oracle/raster_oracle.py:69)
    ang = rng.uniform(0, 3.1, (n, 1))
    boxes = np.concatenate([rng.uniform(8, 120, (n, 2)), np.cos(ang), -np.sin(ang), rng.uniform(1, 6, (n, 2)), rng.random((n, 3))], axis=1)
    outs = set()
    for _ in range(reps):
        a = imageops.resize_normalize(imgs, (64, 48))
        b = imageops.resize_normalize(fimg, (64, 48))
        c = rz.rasterize_boxes(boxes, (128, 128), (0.5, 0.5, 0.5))
        outs.add((float(a.double().abs().sum()).hex(), float(b.double().abs().sum()).hex(), float(c.double().abs().sum()).hex()))
    return outs


@pytest.fixture
def background_load():
    """two other processes looping forwards of the default network (fp32-equivalent and bf16) on the same GPU"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    procs = [subprocess.Popen([sys.executable, os.path.join("tools", "race_probe.py"), "fwd", "DEFAULT3", b, dt, "1000000"],
                              cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
             for b, dt in (("2", "fp32"), ("4", "bf16"))]
    time.sleep(25)   # imports + plans: both are inside their loops by now
    assert all(p.poll() is None for p in procs), "a background loader exited"
    yield procs
    for p in procs:
        p.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()


def test_results_do_not_depend_on_other_processes_on_the_gpu(background_load):
    # (solo references are not needed for the repeated legs: the first repetition is as loaded as the last)
    worst, same = _conv_out_runs(150)
    assert worst <= 2e-5 and same, (worst, same)                    # (was: 1e-3 .. 1e-1 off in ~90 % of the launches)
    assert len(_streaming_rows(60)) == 1
    for cfg, batch, dtype, reps, train in ((configs.CFG1, 4, "fp32", 60, True), (configs.CFG1, 4, "bf16", 60, True),
                                           (configs.CFG1, 4, "fp32", 60, False), (configs.CFG4_SMALL, 2, "fp32", 30, False),
                                           (configs.DEFAULT3, 2, "fp32", 8, True), (configs.DEFAULT3, 2, "bf16", 8, False)):
        sums = _forward_sums(cfg, batch, dtype, reps, train)
        assert len(set(sums)) == 1, (batch, dtype, train, sorted(set(sums))[:4])


def test_training_steps_under_load_are_the_solo_steps():
    cases = ((configs.CFG1, 4, "fp32", 6), (configs.CFG1, 4, "bf16", 6), (configs.CFG4_SMALL, 2, "fp32", 4),
             (configs.DEFAULT3, 2, "fp32", 3))
    solo = [_train_run(*c) for c in cases]
    torch.cuda.synchronize()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    procs = [subprocess.Popen([sys.executable, os.path.join("tools", "race_probe.py"), "fwd", "DEFAULT3", b, dt, "1000000"],
                              cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
             for b, dt in (("2", "fp32"), ("4", "bf16"))]
    try:
        time.sleep(25)
        assert all(p.poll() is None for p in procs), "a background loader exited"
        for c, want in zip(cases, solo):
            for _ in range(2):
                assert _train_run(*c) == want, c[1:]
    finally:
        for p in procs:
            p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
