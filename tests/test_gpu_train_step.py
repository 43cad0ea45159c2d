"""Whole-network training-step parity: given identical (batch, noise, timesteps,
weights, optimizer state) the HIP engine's loss, gradients and AdamW update match the torch-CPU oracle
(training is unseeded in the reference, so per-step math parity is the contract -- App. A.4)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import ops, synth  # noqa: E402
from oracle.scheduler_oracle import OracleDDPMScheduler  # noqa: E402
from oracle.unet_oracle import OracleUNet2DModel  # noqa: E402
from tests.common import CFG1, CFG4_SMALL, max_abs, noisy_inputs, rel_l2, synth_weights  # noqa: E402

DEV = "cuda"


def _grads_close(net, ora, rel=2e-4):
    bad = []
    og = dict(ora.named_parameters())
    for name, p in net.named_parameters():
        g, w = p.grad.detach().cpu(), og[name].grad
        scale = float(w.abs().max()) + 1e-12
        if max_abs(g, w) > 5e-4 * scale + 1e-7 or (float(w.norm()) > 1e-6 and rel_l2(g, w) > rel):
            bad.append((name, rel_l2(g, w), max_abs(g, w), scale))
    assert not bad, bad[:8]


@pytest.mark.parametrize("cfg", [CFG1, CFG4_SMALL], ids=["cfg1_tiny", "attn_blocks"])
def test_training_step_matches_oracle(cfg):
    net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).train()
    ora = synth_weights(OracleUNet2DModel(**cfg)).train()
    sch, osch = d.DDPMScheduler(), OracleDDPMScheduler()
    b = 3
    ss = cfg["sample_size"]
    x0 = torch.from_numpy(synth.synth_scene_rasters(b, cfg["in_channels"], ss, ss, 5))
    noise = torch.from_numpy(synth.normal(6, tuple(x0.shape)))
    t = torch.tensor([12, 500, 987])
    # reference step (training_pipeline.py:72-86)
    noisy_o = osch.add_noise(x0, noise, t)
    pred_o = ora(noisy_o, t, return_dict=False)[0]
    loss_o = F.mse_loss(pred_o, noise)
    loss_o.backward()
    # engine step
    noisy = sch.add_noise(x0.to(DEV), noise.to(DEV), t.to(DEV))
    assert torch.equal(noisy.cpu(), noisy_o)
    pred = net(noisy, t.to(DEV), return_dict=False)[0]
    assert pred.requires_grad
    assert rel_l2(pred.detach().cpu(), pred_o.detach()) <= 1e-4
    loss = d.mse_loss(pred, noise.to(DEV))
    loss.backward()
    assert abs(float(loss.detach().cpu()) - float(loss_o)) <= 1e-5 * float(loss_o)
    _grads_close(net, ora)
    # gradient accumulation: a second backward adds to .grad (accelerator.accumulate semantics)
    pred = net(noisy, t.to(DEV), return_dict=False)[0]
    d.mse_loss(pred, noise.to(DEV)).backward()
    g1 = net.conv_out.weight.grad.detach().cpu()
    assert rel_l2(g1, 2 * ora.conv_out.weight.grad) <= 2e-4


def test_clip_and_adamw_step_match_torch():
    """train.py:66 AdamW(lr) + training_pipeline.py:88 clip_grad_norm_(1.0) on the whole model (flat-slab path),
    two steps.  Adam turns round-off-level gradients (e.g. the mathematically-zero gradient of a conv bias that
    feeds a GroupNorm) into +-lr updates, so the optimizer is compared on IDENTICAL gradients: the engine's own
    backward creates the slab, then the oracle's gradients are copied in (gradient parity itself is the test
    above)."""
    net = synth_weights(d.UNet2DModel(**CFG1)).to(DEV).train()
    ora = synth_weights(OracleUNet2DModel(**CFG1)).train()
    opt = d.AdamW(net.parameters(), lr=1e-3)
    oopt = torch.optim.AdamW(ora.parameters(), lr=1e-3)
    osch = OracleDDPMScheduler()
    for step in range(2):
        x0 = torch.from_numpy(synth.synth_scene_rasters(2, 3, 64, 64, 20 + step))
        noise = torch.from_numpy(synth.normal(30 + step, tuple(x0.shape)))
        t = torch.tensor([100 + step, 900 - step])
        noisy = osch.add_noise(x0, noise, t)
        F.mse_loss(ora(noisy, t, return_dict=False)[0], noise).backward()
        d.mse_loss(net(noisy.to(DEV), t.to(DEV), return_dict=False)[0], noise.to(DEV)).backward()
        _grads_close(net, ora)
        for (_, p), (_, q) in zip(net.named_parameters(), ora.named_parameters()):
            p.grad.copy_(q.grad)
        n_ref = torch.nn.utils.clip_grad_norm_(ora.parameters(), 1.0)
        oopt.step()
        oopt.zero_grad()
        n_got = d.clip_grad_norm_(net.parameters(), 1.0)
        assert abs(float(n_got) - float(n_ref)) <= 1e-5 * float(n_ref), (float(n_got), float(n_ref))
        opt.step()
        opt.zero_grad()
        assert all(float(p.grad.abs().max()) == 0.0 for p in net.parameters())
        worst = max(max_abs(p.detach().cpu(), q.detach()) for (_, p), (_, q) in
                    zip(net.named_parameters(), ora.named_parameters()))
        assert worst <= 2e-6, worst
    # the flat-slab path was taken: one fused launch over all parameters
    assert list(opt._flat.values())[0], "AdamW fell back to per-tensor launches"
    # updated weights are picked up by the next forward (engine-layout copies refresh on the version bump)
    x = noisy_inputs(CFG1, 1)
    with torch.no_grad():
        want = ora(x, 7).sample
        got = net(x.to(DEV), 7).sample
    assert rel_l2(got.cpu(), want) <= 1e-4
