"""Free-running sampling trajectories at the BASELINE configurations' own sizes against the oracle's stored runs
(tests/golden/trajectory_golden.npz, tests/golden/make_trajectory_golden.py) -- SURVEY 8c's last tolerance: "free-running
10/50-step trajectory: rel-L2 <= 1e-3 on the final image; uint8 <= 1 LSB on <= 0.1 % of pixels".

  cfg2_ddim50       BASELINE configs[1]: the bench's own workload, 50-step DDIM of the 256x256x4 default U-Net
  cfg4_ddim100      BASELINE configs[3]: 100-step DDIM of the 6-level 512x512x4 attention network
  default3_ddpm750  the reference's evaluate call (training_pipeline.py:26-32; generation.py:14-20 runs the same loop at
                    batch 5): 750 ancestral DDPM steps, x_T and every step's noise from torch.manual_seed(14555)

The engine runs every step on its own previous output (nothing is teacher-forced); the image at the END is compared, and the
stored checkpoints give the curve along which the two runs drift apart (written to gpurun_out/trajectory_divergence.json
for DESIGN.md).  Reference call shapes: DDPMPipeline.__call__ as generation.py:14-20 and training_pipeline.py:26-39 use it."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from tests.common import TRAJECTORIES, rel_l2, synth_weights, trajectory_golden, trajectory_x_T  # noqa: E402

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _to_u8(x):
    img = (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy()
    return (img * 255).round().astype("uint8")


def _free_run(key):
    """The engine's own run of a stored trajectory: (x_0 on the CPU, [rel-L2 to the oracle at each stored checkpoint])."""
    cfg, kind, steps, stride, every, _ = TRAJECTORIES[key]
    gold = trajectory_golden()
    net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).eval().requires_grad_(False)
    sch = d.DDIMScheduler() if kind == "ddim" else d.DDPMScheduler()
    sch.set_timesteps(steps)
    x_cpu, gen = trajectory_x_T(key)
    x = x_cpu.to(DEV)
    cps = torch.from_numpy(gold[key + "/checkpoints"])
    curve = []
    for i, tt in enumerate(sch.timesteps):
        t = int(tt)
        if i % every == 0:
            curve.append(rel_l2(x[:, :, ::8, ::8].cpu(), cps[i // every]))
        eps = net(x, t).sample
        if kind == "ddim":
            x = sch.step(eps, t, x).prev_sample
        else:   # DDPMPipeline's order of draws: x_T, then one tensor per step with t > 0 (CPU generator, moved)
            noise = torch.randn(tuple(x.shape), generator=gen).to(DEV) if t > 0 else None
            x = sch.step(eps, t, x, variance_noise=noise).prev_sample
    assert len(curve) == cps.shape[0]
    return x.cpu(), curve


def _record(key, entry):
    path = os.path.join(ROOT, "gpurun_out", "trajectory_divergence.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        allv = json.load(open(path))
    except (OSError, ValueError):
        allv = {}
    allv[key] = entry
    json.dump(allv, open(path, "w"), indent=1)


@pytest.mark.parametrize("key", list(TRAJECTORIES))
def test_free_running_trajectory_ends_on_the_oracles_image(key):
    cfg, kind, steps, stride, every, _ = TRAJECTORIES[key]
    gold = trajectory_golden()
    x, curve = _free_run(key)
    assert torch.isfinite(x).all()
    want = torch.from_numpy(gold[key + "/final"])
    err = rel_l2(x[:, :, ::stride, ::stride], want)
    u8, want_u8 = _to_u8(x), gold[key + "/final_u8"]
    diff = np.abs(u8.astype(np.int16) - want_u8.astype(np.int16))
    frac, worst = float((diff > 0).mean()), int(diff.max())
    mom = torch.from_numpy(gold[key + "/final_moments"])
    ms = x.double().pow(2).mean((0, 2, 3))
    _record(key, {"steps": steps, "scheduler": kind, "final_rel_l2": err, "u8_pixels_differing": frac, "u8_max_lsb": worst,
                  "checkpoint_every": every, "rel_l2_at_checkpoints": curve})
    # SURVEY 8c, written here: final image rel-L2 <= 1e-3; uint8 differs by at most 1 LSB, on at most 0.1 % of the pixels
    assert err <= 1e-3, (key, err, curve)
    assert worst <= 1 and frac <= 1e-3, (key, worst, frac)
    assert ((ms - mom[1]).abs() <= 4e-3 * mom[1] + 1e-9).all(), key
    assert curve[0] == 0.0   # both runs start from the same x_T


def test_pipeline_object_reproduces_the_free_run_of_the_evaluate_call():
    """The DDPMPipeline object seeded like training_pipeline.py:26-32 ends on the stored uint8 image too (rounded as
    generation.py's PIL output is; <= 1 LSB on <= 0.1 % of pixels) -- through `output_type='pil'`, the generation.py path."""
    key = "default3_ddpm750"
    cfg = TRAJECTORIES[key][0]
    net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).eval().requires_grad_(False)
    pipe = d.DDPMPipeline(unet=net, scheduler=d.DDPMScheduler())
    img = pipe(num_inference_steps=750, batch_size=1, generator=torch.manual_seed(14555)).images[0]
    got = np.asarray(img)[None]
    want = trajectory_golden()[key + "/final_u8"]
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert got.shape == want.shape and int(diff.max()) <= 1 and float((diff > 0).mean()) <= 1e-3
