"""Shared helpers for the parity tests (configs of BASELINE.json, synthetic weights / inputs)."""
import numpy as np
import torch

from drivescenegen_amd import synth

CFG1 = dict(sample_size=64, in_channels=3, out_channels=3, layers_per_block=2, block_out_channels=(32, 64),
            down_block_types=("DownBlock2D",) * 2, up_block_types=("UpBlock2D",) * 2)
DEFAULT3 = dict(sample_size=(256, 256), in_channels=3, out_channels=3, layers_per_block=2,
                block_out_channels=(64, 128, 256, 512), down_block_types=("DownBlock2D",) * 4,
                up_block_types=("UpBlock2D",) * 4)
CFG2 = dict(DEFAULT3, in_channels=4, out_channels=4)
# BASELINE configs[3] as read in SURVEY 8d, shrunk spatially for parity runs (same block types)
CFG4_SMALL = dict(sample_size=128, in_channels=4, out_channels=4, layers_per_block=1,
                  block_out_channels=(32, 32, 64, 64), down_block_types=("DownBlock2D", "DownBlock2D",
                                                                         "AttnDownBlock2D", "AttnDownBlock2D"),
                  up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D", "UpBlock2D"))


def synth_weights(module, seed=14555):
    """Load the counter-based synthetic weights into any module with the diffusers key set."""
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed).items()}
    module.load_state_dict(sd)
    return module


def noisy_inputs(cfg, batch, seed=14555):
    """x_t-like inputs: scene rasters mixed with unit noise; float32 [B,C,H,W]."""
    ss = cfg["sample_size"]
    h, w = (ss, ss) if isinstance(ss, int) else ss
    x0 = synth.synth_scene_rasters(batch, cfg["in_channels"], h, w, seed)
    nz = synth.normal(seed + 1, x0.shape)
    return torch.from_numpy((0.6 * x0 + 0.8 * nz).astype(np.float32))


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs(a, b):
    return float((a.double() - b.double()).abs().max())
