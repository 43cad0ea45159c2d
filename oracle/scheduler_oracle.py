"""Torch-CPU restatement of diffusers-0.20.0 ``DDPMScheduler`` / ``DDIMScheduler`` and the
``DDPMPipeline.__call__`` loop -- TEST INFRASTRUCTURE (oracle).

Reference call sites: /root/reference/DriveSceneGen/scripts/train.py:65 (``DDPMScheduler()``),
/root/reference/DriveSceneGen/pipeline/training_pipeline.py:76,80 (``num_train_timesteps``, ``add_noise``),
training_pipeline.py:26-32 and /root/reference/DriveSceneGen/scripts/generation.py:14-20 (pipeline call).
Formulas: SURVEY.md Appendix A.3, A.3b, A.4.  Parity unpinned (oracle/__init__.py); known-answer
values of Appendix C are asserted by tests/test_oracle_kat.py.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch


def _randn(shape, generator, dtype=torch.float32):
    return torch.randn(shape, generator=generator, dtype=dtype)


class OracleDDPMScheduler:
    """All-default ``DDPMScheduler()`` (Appendix A.3)."""

    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, clip_sample=True,
                 clip_sample_range=1.0):
        self.num_train_timesteps = num_train_timesteps
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, clip_sample=clip_sample,
                                      clip_sample_range=clip_sample_range)
        self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def add_noise(self, x0, noise, timesteps):
        ac = self.alphas_cumprod.to(device=x0.device, dtype=x0.dtype)
        timesteps = timesteps.to(x0.device)
        sa = ac[timesteps] ** 0.5
        sa = sa.flatten()
        while sa.dim() < x0.dim():
            sa = sa.unsqueeze(-1)
        sb = (1 - ac[timesteps]) ** 0.5
        sb = sb.flatten()
        while sb.dim() < x0.dim():
            sb = sb.unsqueeze(-1)
        return sa * x0 + sb * noise

    def _prev(self, t):
        n = self.num_inference_steps if self.num_inference_steps else self.num_train_timesteps
        return t - self.num_train_timesteps // n

    def step(self, model_output, timestep, sample, generator=None, noise=None):
        t = int(timestep)
        prev_t = self._prev(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        c0 = (a_prev ** 0.5 * cur_beta) / b_t
        ct = cur_alpha ** 0.5 * b_prev / b_t
        prev = c0 * x0 + ct * sample
        if t > 0:
            z = noise if noise is not None else _randn(model_output.shape, generator, model_output.dtype)
            var = torch.clamp(b_prev / b_t * cur_beta, min=1e-20)
            prev = prev + (var ** 0.5) * z
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


class OracleDDIMScheduler(OracleDDPMScheduler):
    """``DDIMScheduler`` 0.20.0 defaults (Appendix A.3b): eta=0, clip, set_alpha_to_one, leading."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.final_alpha_cumprod = torch.tensor(1.0)

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, noise=None):
        t = int(timestep)
        prev_t = self._prev(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        b_prev = 1 - a_prev
        variance = (b_prev / b_t) * (1 - a_t / a_prev)
        std = eta * variance ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * model_output
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            z = noise if noise is not None else _randn(model_output.shape, generator, model_output.dtype)
            prev = prev + std * z
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


def oracle_pipeline(unet, scheduler, batch_size=1, generator=None, num_inference_steps=1000,
                    output_type="pil", ddim=False, trajectory=None):
    """``DDPMPipeline.__call__`` (Appendix A.4): CPU-generator noise, loop, [0,1] HWC float post-process.

    ``trajectory``: optional list receiving (t, eps, x_prev) per step for teacher-forced parity tests.
    Returns the float ndarray [B,H,W,C] (or uint8 with rounding when ``output_type == "pil"``).
    """
    ss = unet.config.sample_size
    if isinstance(ss, int):
        shape = (batch_size, unet.config.in_channels, ss, ss)
    else:
        shape = (batch_size, unet.config.in_channels, *ss)
    image = _randn(shape, generator)
    scheduler.set_timesteps(num_inference_steps)
    with torch.no_grad():
        for t in scheduler.timesteps:
            eps = unet(image, t).sample
            if ddim:
                image = scheduler.step(eps, t, image, eta=0.0, generator=generator).prev_sample
            else:
                image = scheduler.step(eps, t, image, generator=generator).prev_sample
            if trajectory is not None:
                trajectory.append((int(t), eps.clone(), image.clone()))
    image = (image / 2 + 0.5).clamp(0, 1)
    image = image.cpu().permute(0, 2, 3, 1).numpy()
    if output_type == "pil":
        return (image * 255).round().astype("uint8")
    return image


def cosine_lr_lambda(step: int, warmup: int, total: int, num_cycles: float = 0.5) -> float:
    """``get_cosine_schedule_with_warmup`` lambda (Appendix A.6; train.py:67-71)."""
    import math
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
