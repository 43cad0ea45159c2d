#!/usr/bin/env python3
"""Regenerate tests/golden/fullsize_golden.npz from the CPU oracle (run from the repo root, ~2 minutes on 8 cores):

    python tests/golden/make_fullsize_golden.py

Expected OUTPUTS of the fp32 torch-CPU oracle at the BASELINE configurations' own network sizes, so that the `-m gpu`
suite does not have to re-run the 56-66 M-parameter oracle on the GPU box's (shared, slow) host cores for every case
(the suite took 848 s of the driver's 1200 s, most of it oracle time).  Inputs and weights are the
deterministic numpy streams of drivescenegen_amd/synth.py, so only outputs are stored; `tests/common.py::fullsize_case`
rebuilds the matching inputs.  Like cfg1_golden.npz these vectors pin the oracle against drift, not against diffusers
(the reference cannot be imported here: parity unpinned, see oracle/__init__.py).

Stored per forward case: the output -- whole for the 256x256 nets, every `stride`-th pixel in both directions for the
larger ones -- plus per-channel fp64 (mean, mean square) of the WHOLE output.  Per training case: the loss, every
gradient tensor's fp64 L2 norm and its entries [::stride] (<= 512 per tensor).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.scheduler_oracle import OracleDDPMScheduler  # noqa: E402
from oracle.unet_oracle import OracleUNet2DModel  # noqa: E402
from tests.common import FULLSIZE_FWD, FULLSIZE_TRAIN, fullsize_case, fullsize_train_case, grad_sample_stride, synth_weights  # noqa: E402


def main():
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    out = {}
    nets = {}

    def net_of(cfg_name, cfg):
        if cfg_name not in nets:
            nets.clear()  # one 56-66 M-parameter oracle at a time
            nets[cfg_name] = synth_weights(OracleUNet2DModel(**cfg)).eval()
        return nets[cfg_name]

    for key in sorted(FULLSIZE_FWD, key=lambda k: FULLSIZE_FWD[k][0]):
        cfg_name, cfg, x, t, stride = fullsize_case(key)
        with torch.no_grad():
            y = net_of(cfg_name, cfg)(x, t).sample
        out[key] = y[:, :, ::stride, ::stride].contiguous().numpy()
        out[key + "/moments"] = torch.stack([y.double().mean((0, 2, 3)), y.double().pow(2).mean((0, 2, 3))]).numpy()
        print(key, tuple(y.shape), "->", out[key].shape, flush=True)
    nets.clear()
    for key in FULLSIZE_TRAIN:
        cfg, x0, noise, t = fullsize_train_case(key)
        ora = synth_weights(OracleUNet2DModel(**cfg)).train()
        noisy = OracleDDPMScheduler().add_noise(x0, noise, t)
        loss = F.mse_loss(ora(noisy, t, return_dict=False)[0], noise)
        loss.backward()
        out[key + "/loss"] = np.array([float(loss.detach())], np.float64)
        norms, samples = [], []
        for name, p in ora.named_parameters():
            g = p.grad.detach().flatten()
            norms.append(float(g.double().norm()))
            samples.append(g[::grad_sample_stride(g.numel())].clone().numpy())
        out[key + "/grad_norms"] = np.array(norms, np.float64)
        out[key + "/grad_samples"] = np.concatenate(samples)
        print(key, "loss", float(loss.detach()), "tensors", len(norms), "samples", out[key + "/grad_samples"].shape, flush=True)
        del ora
    path = os.path.join(ROOT, "tests", "golden", "fullsize_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
