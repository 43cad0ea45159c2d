"""The U-Net configurations BASELINE.json names, as constructor kwargs of ``UNet2DModel``, and the synthetic
weights / inputs benchmarks and parity tests share (there is no network for checkpoints or the dataset).

Reference: the network of /root/reference/DriveSceneGen/scripts/train.py:39-57 (``DEFAULT3``); the other
entries are BASELINE.json's variations of it as SURVEY.md section 8d reads them.
"""
from __future__ import annotations

import numpy as np
import torch

from . import synth

# configs[0]: 64x64x3 raster, tiny net (2 down / 2 up blocks, 32 base channels): the CPU-runnable plumbing case
CFG1 = dict(sample_size=64, in_channels=3, out_channels=3, layers_per_block=2, block_out_channels=(32, 64),
            down_block_types=("DownBlock2D",) * 2, up_block_types=("UpBlock2D",) * 2)
# train.py:39-57 verbatim (3 channels)
DEFAULT3 = dict(sample_size=(256, 256), in_channels=3, out_channels=3, layers_per_block=2,
                block_out_channels=(64, 128, 256, 512), down_block_types=("DownBlock2D",) * 4,
                up_block_types=("UpBlock2D",) * 4)
# configs[1] / configs[2]: 256x256x4 BEV raster, default net
CFG2 = dict(DEFAULT3, in_channels=4, out_channels=4)
CFG3 = CFG2
# configs[3]: 512x512x4, attention at 32^2 and 16^2 (SURVEY 8d reading: 6 levels, 66,294,660 parameters)
CFG4 = dict(sample_size=(512, 512), in_channels=4, out_channels=4, layers_per_block=2,
            block_out_channels=(64, 64, 128, 128, 256, 512),
            down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D",) * 2,
            up_block_types=("AttnUpBlock2D",) * 2 + ("UpBlock2D",) * 4)
# the same block types, shrunk for parity runs that the CPU oracle finishes in seconds
CFG4_SMALL = dict(sample_size=128, in_channels=4, out_channels=4, layers_per_block=1,
                  block_out_channels=(32, 32, 64, 64), down_block_types=("DownBlock2D", "DownBlock2D",
                                                                         "AttnDownBlock2D", "AttnDownBlock2D"),
                  up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D", "UpBlock2D"))
# configs[4]: 256x256x8 map + agent raster, default net, mixed bf16
CFG5 = dict(DEFAULT3, in_channels=8, out_channels=8)

PARAM_COUNTS = {"CFG1": 919_043, "DEFAULT3": 56_574_595, "CFG2": 56_575_748, "CFG4": 66_294_660, "CFG5": 56_580_360}


def synth_weights(module, seed=14555):
    """Load the counter-based synthetic weights (synth.synth_state_dict) into any module with the diffusers key set."""
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed).items()}
    module.load_state_dict(sd)
    return module


def noisy_inputs(cfg, batch, seed=14555):
    """x_t-like inputs: synthetic scene rasters mixed with unit noise; float32 [B, C, H, W]."""
    ss = cfg["sample_size"]
    h, w = (ss, ss) if isinstance(ss, int) else ss
    x0 = synth.synth_scene_rasters(batch, cfg["in_channels"], h, w, seed)
    nz = synth.normal(seed + 1, x0.shape)
    return torch.from_numpy((0.6 * x0 + 0.8 * nz).astype(np.float32))
