#!/usr/bin/env python3
"""Regenerate tests/golden/trajectory_golden.npz from the CPU oracle (run from the repo root, ~25 minutes on 8 cores):

    python tests/golden/make_trajectory_golden.py [case ...]

FREE-RUNNING sampling trajectories of the fp32 torch-CPU oracle at the BASELINE configurations' own network sizes: the
image a whole run produces, which is what SURVEY 8c's last tolerance is about ("free-running 10/50-step trajectory:
rel-L2 <= 1e-3 on the final image; uint8 <= 1 LSB on <= 0.1 % of pixels") and what no single-forward vector can show.

  cfg2_ddim50     BASELINE configs[1]: 50-step DDIM (eta = 0) of the 256x256x4 default U-Net, row 0 of the batch-16 x_T
  cfg4_ddim100    BASELINE configs[3]: 100-step DDIM of the 6-level 512x512x4 network, row 0 of the batch-8 x_T
  default3_ddpm750  the reference's own evaluate call (training_pipeline.py:26-32): 750-step ancestral DDPM, batch 1,
                  x_T and every step's noise from torch.manual_seed(14555) in the order DDPMPipeline draws them
  cfg2_ddim50_c, cfg4_ddim100_c   the two DDIM runs on the CONTRACTIVE weight set (tests/common.py: trajectory_weights --
                  conv2 / to_out.0 of every block scaled by 0.1): two fp32 runs stay together to the end, so the final image
                  can be asked for to SURVEY 8c's letter

Inputs and weights are the deterministic streams of drivescenegen_amd/synth.py (tests/common.py rebuilds them), so only
outputs are stored.  Per case: `final` = x_0 at every `stride`-th pixel (fp32), `final_moments` = per-channel fp64 (mean,
mean square) of the whole x_0, `final_u8` = the whole post-processed image as DDPMPipeline.numpy_to_pil makes it
((x / 2 + 0.5).clamp(0, 1) -> HWC -> * 255 -> round -> uint8; generation.py:17-20), and `checkpoints` = x_t at every
`every`-th step at every 8th pixel -- the curve along which an engine run may drift from the oracle's.

`--sensitivity` adds `self_divergence` (and `self_u8_frac`: the fraction of uint8 values the two runs round differently): the same oracle run again from an x_T moved by 1e-6 (relative), compared with its own
unperturbed run at the same checkpoints and at the end.  With the synthetic (untrained, random) weights these networks are
not contractive: two fp32 runs that differ by one forward's round-off drift apart exponentially, and the free-running
tolerance of SURVEY 8c can only be asked for inside the horizon where the oracle agrees with ITSELF.  `--early` stores that
horizon at full resolution: x_t after each of the first 5 steps, and the perturbed run's distance from it.

Like the other golden files these vectors pin the oracle against drift, not against diffusers (parity unpinned: the
reference's diffusers 0.20.0 cannot be imported here, see oracle/__init__.py).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.scheduler_oracle import OracleDDIMScheduler, OracleDDPMScheduler  # noqa: E402
from oracle.unet_oracle import OracleUNet2DModel  # noqa: E402
from tests.common import TRAJECTORIES, trajectory_weights, trajectory_x_T  # noqa: E402

PATH = os.path.join(ROOT, "tests", "golden", "trajectory_golden.npz")


def to_u8(x):
    img = (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    return (img * 255).round().astype("uint8")


PERTURB = 1e-6   # relative size of the x_T perturbation of the sensitivity runs: the class of one forward's fp32 round-off


EARLY = 5   # the first steps, stored one by one: the stretch where two fp32 runs still agree to the free-running tolerance


def run(key, perturb=0.0, early=False):
    cfg, kind, steps, stride, every = TRAJECTORIES[key][:5]
    net = trajectory_weights(OracleUNet2DModel(**cfg), key).eval()
    sch = OracleDDIMScheduler() if kind == "ddim" else OracleDDPMScheduler()
    sch.set_timesteps(steps)
    x, gen = trajectory_x_T(key)
    if perturb:   # the SAME oracle from an x_T moved by `perturb` (relative, seeded): how fast two fp32 runs drift apart
        from drivescenegen_amd import synth
        x = x + perturb * x.pow(2).mean().sqrt() * torch.from_numpy(synth.normal(999, tuple(x.shape)))
    cps = []
    t0 = time.time()
    with torch.no_grad():
        for i, tt in enumerate(sch.timesteps):
            t = int(tt)
            if early and i == EARLY:
                return np.stack(cps[1:] + [x[:, :, ::8, ::8].clone().numpy()])   # x after steps 1 .. EARLY
            if early or i % every == 0:
                cps.append(x[:, :, ::8, ::8].clone().numpy())
            eps = net(x, t).sample
            if kind == "ddim":
                x = sch.step(eps, t, x).prev_sample
            else:
                noise = torch.randn(tuple(x.shape), generator=gen) if t > 0 else None
                x = sch.step(eps, t, x, noise=noise).prev_sample
            if i % 25 == 0:
                print(key, "step", i, "t", t, f"{time.time() - t0:.0f}s", flush=True)
    assert torch.isfinite(x).all()
    if perturb:
        return np.stack(cps), x[:, :, ::stride, ::stride].contiguous().numpy()
    return {key + "/final": x[:, :, ::stride, ::stride].contiguous().numpy(),
            key + "/final_moments": torch.stack([x.double().mean((0, 2, 3)), x.double().pow(2).mean((0, 2, 3))]).numpy(),
            key + "/final_u8": to_u8(x), key + "/checkpoints": np.stack(cps)}


def rel(a, b):
    a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    out = dict(np.load(PATH)) if os.path.exists(PATH) else {}
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--early" in sys.argv:
        # key/early = x_t after each of the first EARLY steps (every 8th pixel); key/early_self_divergence = the perturbed
        # oracle run against it, step by step
        for key in (args or list(TRAJECTORIES)):
            base, pert = run(key, 0.0, True), run(key, PERTURB, True)
            out[key + "/early"] = base
            out[key + "/early_self_divergence"] = np.array([rel(a, b) for a, b in zip(pert, base)])
            np.savez_compressed(PATH, **out)
            print("early", key, out[key + "/early_self_divergence"], flush=True)
        return
    if "--sensitivity" in sys.argv:
        # key/self_divergence = [rel-L2 between the oracle's run and the oracle's run from the perturbed x_T at every stored
        # checkpoint ..., at the final image]: the envelope any other fp32 implementation's drift is judged against
        for key in (args or list(TRAJECTORIES)):
            cps, fin = run(key, PERTURB)
            out[key + "/self_divergence"] = np.array([rel(c, g) for c, g in zip(cps, out[key + "/checkpoints"])] +
                                                     [rel(fin, out[key + "/final"])])
            # ... and what that drift does to the uint8 image (on the stored, strided pixels): the fraction of values the
            # oracle's two runs round differently -- no implementation can be asked for fewer than the oracle gives itself
            u8 = lambda a: (np.clip(a / 2 + 0.5, 0, 1) * 255).round().astype("uint8")
            out[key + "/self_u8_frac"] = np.array([float((u8(fin) != u8(out[key + "/final"])).mean())])
            np.savez_compressed(PATH, **out)
            print("sensitivity", key, out[key + "/self_divergence"], flush=True)
        return
    for key in (args or list(TRAJECTORIES)):
        out.update(run(key))
        np.savez_compressed(PATH, **out)
        print("wrote", key, "->", PATH, os.path.getsize(PATH), "bytes", flush=True)


if __name__ == "__main__":
    main()
