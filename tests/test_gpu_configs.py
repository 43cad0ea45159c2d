"""Every BASELINE.json config at its own size (configs_untested): configs[2] -- a training step of the
256x256x4 default network against torch-CPU autograd of the oracle plus a batch-64 size-independent property --,
configs[3] -- the 6-level 512x512 attention network (SURVEY 8d reading, 66,294,660 parameters) --, configs[4] -- a bf16
training step of the 8-channel network --, and the reference's own sampling call (750-step seeded DDPM at batch 1,
training_pipeline.py:26-32) teacher-forced against the oracle every 50 steps."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import synth  # noqa: E402
from oracle.scheduler_oracle import OracleDDPMScheduler  # noqa: E402
from oracle.unet_oracle import OracleUNet2DModel  # noqa: E402
from tests.common import CFG3, CFG4, CFG5, DEFAULT3, PARAM_COUNTS, max_abs, noisy_inputs, rel_l2, synth_weights  # noqa: E402

DEV = "cuda"


def _train_inputs(cfg, b, seed=5):
    ss = cfg["sample_size"]
    h, w = (ss, ss) if isinstance(ss, int) else ss
    x0 = torch.from_numpy(synth.synth_scene_rasters(b, cfg["in_channels"], h, w, seed))
    noise = torch.from_numpy(synth.normal(seed + 1, tuple(x0.shape)))
    t = torch.from_numpy((synth.uniform01(seed + 2, b) * 1000).astype(np.int64))
    return x0, noise, t


def _oracle_grads(cfg, x0, noise, t):
    ora = synth_weights(OracleUNet2DModel(**cfg)).train()
    noisy = OracleDDPMScheduler().add_noise(x0, noise, t)
    loss = F.mse_loss(ora(noisy, t, return_dict=False)[0], noise)
    loss.backward()
    return ora, noisy, float(loss.detach())


def test_cfg3_training_step_256_vs_oracle_autograd():
    """configs[2] network (256x256x4, 56,575,748 parameters): loss and all 282 gradients of one DDPM training step at
    batch 2 against torch-CPU autograd of the fp32 oracle."""
    x0, noise, t = _train_inputs(CFG3, 2)
    ora, noisy, loss_o = _oracle_grads(CFG3, x0, noise, t)
    net = synth_weights(d.UNet2DModel(**CFG3)).to(DEV).train()
    assert sum(p.numel() for p in net.parameters()) == PARAM_COUNTS["CFG2"] and len(list(net.parameters())) == 282
    loss = d.mse_loss(net(noisy.to(DEV), t.to(DEV), return_dict=False)[0], noise.to(DEV))
    loss.backward()
    assert abs(float(loss.detach().cpu()) - loss_o) <= 2e-5 * loss_o
    og = dict(ora.named_parameters())
    bad, num, den = [], 0.0, 0.0
    for name, p in net.named_parameters():
        g, w = p.grad.detach().cpu(), og[name].grad
        num += float((g.double() - w.double()).pow(2).sum())
        den += float(w.double().pow(2).sum())
        scale = float(w.abs().max()) + 1e-12
        if max_abs(g, w) > 1e-3 * scale + 1e-8 or (float(w.norm()) > 1e-7 and rel_l2(g, w) > 5e-4):
            bad.append((name, rel_l2(g, w), max_abs(g, w), scale))
    assert (num / den) ** 0.5 <= 1e-4, (num / den) ** 0.5
    assert not bad, bad[:8]


def test_cfg3_batch64_gradients_equal_those_of_the_repeated_pair():
    """configs[2] at its own batch (64 per GPU): a mean-reduced loss over 32 copies of a 2-sample batch has the
    gradients of the 2-sample batch -- a size-independent check of the full-size backward (split-K over 64 images,
    bucket / slab arithmetic), finite everywhere, within the memory of one MI355X."""
    x0, noise, t = _train_inputs(CFG3, 2, seed=9)
    sch = d.DDPMScheduler()
    net = synth_weights(d.UNet2DModel(**CFG3)).to(DEV).train()
    noisy = sch.add_noise(x0.to(DEV), noise.to(DEV), t.to(DEV))
    d.mse_loss(net(noisy, t.to(DEV), return_dict=False)[0], noise.to(DEV)).backward()
    small = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    for p in net.parameters():
        p.grad.zero_()
    torch.cuda.reset_peak_memory_stats()
    rep = lambda a: a.to(DEV).repeat(32, *([1] * (a.dim() - 1)))
    loss = d.mse_loss(net(rep(noisy), rep(t), return_dict=False)[0], rep(noise))
    loss.backward()
    assert torch.isfinite(loss.detach()).all()
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    assert peak < 200, peak
    worst = 0.0
    for n, p in net.named_parameters():
        assert torch.isfinite(p.grad).all(), n
        if float(small[n].norm()) > 1e-7:
            worst = max(worst, rel_l2(p.grad.cpu(), small[n].cpu()))
    assert worst <= 5e-4, worst


def test_cfg4_six_level_512_forward_vs_oracle_and_row_independence():
    """configs[3]: 512x512x4, (64,64,128,128,256,512), attention at 32^2 and 16^2 (66,294,660 parameters): batch-1
    forward against the CPU oracle, and at the config's batch 8 every row equals its own batch-1 run bit for bit."""
    net = synth_weights(d.UNet2DModel(**CFG4)).to(DEV).eval().requires_grad_(False)
    assert sum(p.numel() for p in net.parameters()) == PARAM_COUNTS["CFG4"]
    from tests.common import same_kernels_at_any_batch
    x = noisy_inputs(CFG4, 8)
    t = torch.tensor([990, 700, 500, 300, 100, 50, 10, 0])
    with same_kernels_at_any_batch():   # (its 16^2 levels are small grids even at batch 8: K would be cut by batch size)
        got = net(x.to(DEV), t.to(DEV)).sample
    fast = net(x.to(DEV), t.to(DEV)).sample
    assert rel_l2(fast.cpu(), got.cpu()) <= 2e-6
    assert torch.isfinite(got).all()
    ora = synth_weights(OracleUNet2DModel(**CFG4)).eval()
    with torch.no_grad():
        want = ora(x[:1], t[:1]).sample
    assert rel_l2(got[:1].cpu(), want) <= 1e-4 and max_abs(got[:1].cpu(), want) <= 2e-4 * max(1.0, float(want.abs().max()))
    for i in (3, 7):
        with same_kernels_at_any_batch():
            assert torch.equal(net(x[i:i + 1].to(DEV), t[i:i + 1].to(DEV)).sample, got[i:i + 1])
        assert rel_l2(net(x[i:i + 1].to(DEV), t[i:i + 1].to(DEV)).sample.cpu(), got[i:i + 1].cpu()) <= 2e-6


def test_cfg5_bf16_training_step_256_vs_oracle_autograd():
    """configs[4] network (256x256x8, 56,580,360 parameters), one mixed-bf16 training step at batch 2 against fp32
    autograd of the oracle: loss within 1e-2, the whole gradient vector within 3e-2 (rel-L2)."""
    x0, noise, t = _train_inputs(CFG5, 2, seed=21)
    ora, noisy, loss_o = _oracle_grads(CFG5, x0, noise, t)
    net = synth_weights(d.UNet2DModel(**CFG5)).to(DEV).train().set_compute_dtype("bf16")
    loss = d.mse_loss(net(noisy.to(DEV), t.to(DEV), return_dict=False)[0], noise.to(DEV))
    loss.backward()
    assert abs(float(loss.detach().cpu()) - loss_o) <= 1e-2 * loss_o
    og = dict(ora.named_parameters())
    num = sum(float((p.grad.detach().cpu().double() - og[n].grad.double()).pow(2).sum()) for n, p in net.named_parameters())
    den = sum(float(w.grad.double().pow(2).sum()) for w in og.values())
    assert (num / den) ** 0.5 <= 3e-2, (num / den) ** 0.5


def test_reference_evaluate_call_750_steps_teacher_forced():
    """training_pipeline.py:26-32: 750-step DDPM, batch 1, CPU generator seeded 14555, on the train.py:39-57 network.
    The engine free-runs the whole trajectory; every 50th step its x_t is handed to the CPU oracle, whose eps and
    x_{t-1} for that step must agree (eps rel-L2 <= 1e-4, x_{t-1} <= 1e-4): 15 teacher-forced checkpoints over the
    749 ... 0 timestep table.  The pipeline object, seeded the same way, reproduces the free-running result bit for bit."""
    net = synth_weights(d.UNet2DModel(**DEFAULT3)).to(DEV).eval().requires_grad_(False)
    ora = synth_weights(OracleUNet2DModel(**DEFAULT3)).eval()
    sch, osch = d.DDPMScheduler(), OracleDDPMScheduler()
    sch.set_timesteps(750)
    osch.set_timesteps(750)
    assert [int(v) for v in sch.timesteps[:3]] == [749, 748, 747] and int(sch.timesteps[-1]) == 0
    gen = torch.manual_seed(14555)
    x = torch.randn((1, 3, 256, 256), generator=gen).to(DEV)
    checked = 0
    for i, tt in enumerate(sch.timesteps):
        t = int(tt)
        eps = net(x, t).sample
        noise = torch.randn((1, 3, 256, 256), generator=gen) if t > 0 else None
        nxt = sch.step(eps, t, x, variance_noise=None if noise is None else noise.to(DEV)).prev_sample
        if i % 50 == 0 or t == 0:
            with torch.no_grad():
                oeps = ora(x.cpu(), t).sample
            onxt = osch.step(oeps, t, x.cpu(), noise=noise).prev_sample
            assert rel_l2(eps.cpu(), oeps) <= 1e-4, (t, rel_l2(eps.cpu(), oeps))
            assert rel_l2(nxt.cpu(), onxt) <= 1e-4, (t, rel_l2(nxt.cpu(), onxt))
            checked += 1
        x = nxt
    assert checked == 16 and torch.isfinite(x).all()
    pipe = d.DDPMPipeline(unet=net, scheduler=d.DDPMScheduler())
    img = pipe(num_inference_steps=750, batch_size=1, generator=torch.manual_seed(14555), output_type="np.array",
               return_dict=False)[0]
    want = (x.cpu() / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    assert img.shape == (1, 256, 256, 3) and np.array_equal(img, want)
