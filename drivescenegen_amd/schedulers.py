"""``DDPMScheduler`` / ``DDIMScheduler`` with the diffusers-0.20.0 protocol DriveSceneGen uses.

Reference call sites: /root/reference/DriveSceneGen/scripts/train.py:65 (``DDPMScheduler()``, all defaults),
training_pipeline.py:76 (``.num_train_timesteps`` as a direct attribute), training_pipeline.py:80 and
train.py:91 (``add_noise``), and ``set_timesteps`` / ``step`` inside the DDPMPipeline loop
(training_pipeline.py:26-32, generation.py:14-20).  DDIM is the BASELINE.json extension (configs[1], [3]).
Formulas: SURVEY.md App. A.3 / A.3b.

Host side (this file): the beta / alpha-bar tables, the integer timestep tables and the per-step fp32
scalars, computed with the same fp32 operation order as the reference so they are bit-identical.
Device side: the elementwise tensor math, in libdsg.so (dsg_add_noise / dsg_ddpm_step / dsg_ddim_step).
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from .unet import FrozenConfig


class SchedulerOutput(SimpleNamespace):
    pass


def _randn_like_reference(shape, generator, device, dtype):
    """diffusers ``randn_tensor``: a CPU generator samples on CPU and the result is moved (App. A.4)."""
    if generator is not None and generator.device.type == "cpu" and torch.device(device).type != "cpu":
        return torch.randn(shape, generator=generator, device="cpu", dtype=dtype).to(device)
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)


class HostNoise:
    """A step's variance noise in PINNED host memory, read in place by the step kernel (one pass over PCIe) instead of being
    copied to the device first -- what ``DDPMPipeline`` hands to ``step(..., variance_noise=...)`` when the caller's generator
    lives on the CPU (training_pipeline.py:26-32).  `consumed` is recorded on the launch stream behind the kernel that read
    the buffer; its owner waits for it before writing the buffer again."""

    def __init__(self, tensor):
        if tensor.is_cuda or not tensor.is_contiguous() or tensor.dtype != torch.float32 or not tensor.is_pinned():
            raise ValueError("HostNoise: a contiguous pinned fp32 host tensor is required")
        self.tensor = tensor
        self.consumed = None
        # the address the KERNEL reads: asked of the runtime, not assumed equal to the host address (torch's host-register
        # configuration of the pinned allocator maps the two apart)
        import ctypes
        dev = ctypes.c_void_p()
        _lib.check(_lib.load().dsg_host_device_pointer(tensor.data_ptr(), ctypes.byref(dev)))
        self.device_ptr = dev.value

    @property
    def shape(self):
        return self.tensor.shape

    def wait_consumed(self):
        if self.consumed is not None:
            self.consumed.synchronize()


class DDPMScheduler:
    config_name = "scheduler_config.json"
    _class_name = "DDPMScheduler"
    order = 1

    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                     thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0,
                     sample_max_value=1.0, timestep_spacing="leading", steps_offset=0)

    def __init__(self, **kwargs):
        cfg = dict(self._defaults)
        unknown = set(kwargs) - set(cfg)
        if unknown:
            raise TypeError(f"{self._class_name}: unexpected arguments {sorted(unknown)}")
        cfg.update(kwargs)
        for key in ("beta_schedule", "trained_betas", "prediction_type", "thresholding", "timestep_spacing"):
            if cfg[key] != self._defaults[key]:
                raise NotImplementedError(f"{self._class_name}: {key}={cfg[key]!r} is outside the DriveSceneGen "
                                          f"path (supported: {self._defaults[key]!r})")
        self._check_extra(cfg)
        self.config = FrozenConfig(**cfg)
        n = cfg["num_train_timesteps"]
        self.betas = torch.linspace(cfg["beta_start"], cfg["beta_end"], n, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.custom_timesteps = False
        self.timesteps = torch.from_numpy(np.arange(0, n)[::-1].copy())
        self._dev_tables = {}
        self._scalar_cache = {}

    def _check_extra(self, cfg):
        if cfg["variance_type"] != "fixed_small":
            raise NotImplementedError("DDPMScheduler: only variance_type='fixed_small' (the reference default)")

    # training_pipeline.py:76 reads this attribute directly
    @property
    def num_train_timesteps(self):
        return self.config.num_train_timesteps

    def __len__(self):
        return self.config.num_train_timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError(f"num_inference_steps {num_inference_steps} > num_train_timesteps {n_train}")
        self.num_inference_steps = num_inference_steps
        ratio = n_train // num_inference_steps  # integer floor: 1000 // 750 == 1 (SURVEY headline finding 4)
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def previous_timestep(self, t: int) -> int:
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return t - self.config.num_train_timesteps // n

    # ---- add_noise (training_pipeline.py:80; train.py:91 with an HWC image and timesteps=[1]) ----
    def _sqrt_tables(self, device):
        key = str(device)
        if key not in self._dev_tables:
            ac = self.alphas_cumprod
            self._dev_tables[key] = ((ac ** 0.5).to(device), ((1 - ac) ** 0.5).to(device))
        return self._dev_tables[key]

    def add_noise(self, original_samples, noise, timesteps):
        x0 = original_samples
        if not x0.is_cuda:
            raise RuntimeError("DDPMScheduler.add_noise runs on the MI355X HIP engine only (got a CPU tensor)")
        if x0.dtype != torch.float32:
            raise RuntimeError("add_noise: fp32 only")
        sa_t, sb_t = self._sqrt_tables(x0.device)
        t = timesteps.to(x0.device).flatten()
        sa, sb = sa_t[t].contiguous(), sb_t[t].contiguous()
        n = t.numel()
        if x0.dim() == 0 or (n != 1 and n != x0.shape[0]):
            raise ValueError("add_noise: timesteps must have one entry per leading-dim sample (or one entry)")
        if n == 1:
            per = x0.numel()
        else:
            per = x0.numel() // n
        x0c, nz = x0.contiguous(), noise.to(x0.device, x0.dtype).contiguous()
        out = torch.empty_like(x0c)
        with torch.cuda.device(x0.device):
            _lib.check(_lib.load().dsg_add_noise(_lib.ptr(x0c), _lib.ptr(nz), _lib.ptr(sa), _lib.ptr(sb),
                                                _lib.ptr(out), n, per, _lib.stream_ptr(x0.device)))
        return out

    def add_noise_device(self, original_samples, timesteps, seed: int, offset: int):
        """(noisy, noise): the training step's ``noise = torch.randn(shape).to(device)`` (training_pipeline.py:72) and
        ``add_noise(x0, noise, t)`` (:80) as ONE pass of a library kernel -- the noise is a counter-based Philox4x32-10 /
        Box-Muller stream named by (seed, offset) (``dsg_add_noise_philox``, include/dsg.h), not the host generator's:
        opt-in (``train_steps(..., noise="device")``), for throughput.  `noisy` is bitwise ``add_noise(x0, noise, t)``."""
        x0 = original_samples
        if not x0.is_cuda or x0.dtype != torch.float32:
            raise RuntimeError("DDPMScheduler.add_noise_device runs on the MI355X HIP engine only (an fp32 GPU tensor)")
        sa_t, sb_t = self._sqrt_tables(x0.device)
        t = timesteps.to(x0.device).flatten()
        n = t.numel()
        if x0.dim() == 0 or (n != 1 and n != x0.shape[0]):
            raise ValueError("add_noise_device: timesteps must have one entry per leading-dim sample (or one entry)")
        sa, sb = sa_t[t].contiguous(), sb_t[t].contiguous()
        x0c = x0.contiguous()
        noisy, noise = torch.empty_like(x0c), torch.empty_like(x0c)
        with torch.cuda.device(x0.device):
            _lib.check(_lib.load().dsg_add_noise_philox(_lib.ptr(x0c), _lib.ptr(sa), _lib.ptr(sb), _lib.ptr(noisy),
                                                       _lib.ptr(noise), n, x0c.numel() // n, int(seed) & (2 ** 64 - 1),
                                                       int(offset) & (2 ** 64 - 1), _lib.stream_ptr(x0.device)))
        return noisy, noise

    # ---- reverse step -----------------------------------------------------------------------------
    def step_scalars(self, t: int):
        """fp32 scalars of DDPMScheduler.step for timestep t, in the reference's operation order (memoised per
        (t, previous timestep): a dozen 0-d tensor operations, ~50 us of host time per denoising step otherwise)."""
        key = (t, self.previous_timestep(t))
        hit = self._scalar_cache.get(key)
        if hit is None:
            hit = self._scalar_cache[key] = self._step_scalars(t)
        return hit

    def _step_scalars(self, t: int):
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        c0 = (a_prev ** 0.5 * cur_beta) / b_t
        ct = cur_alpha ** 0.5 * b_prev / b_t
        var = torch.clamp(b_prev / b_t * cur_beta, min=1e-20)
        return dict(sqrt_beta_prod_t=float(b_t ** 0.5), sqrt_alpha_prod_t=float(a_t ** 0.5), coef_x0=float(c0),
                    coef_xt=float(ct), sigma=float(var ** 0.5))

    def step(self, model_output, timestep, sample, generator=None, return_dict: bool = True, variance_noise=None):
        if not sample.is_cuda:
            raise RuntimeError("DDPMScheduler.step runs on the MI355X HIP engine only (got a CPU tensor)")
        t = int(timestep)
        s = self.step_scalars(t)
        noise, host, nptr = None, None, None
        if t > 0:
            if isinstance(variance_noise, HostNoise):    # pinned host buffer: the kernel reads it in place
                host = variance_noise
                if tuple(host.shape) != tuple(sample.shape):
                    raise ValueError(f"variance_noise has shape {tuple(host.shape)}, the sample {tuple(sample.shape)}")
                nptr = host.device_ptr
            else:
                noise = variance_noise if variance_noise is not None else _randn_like_reference(
                    model_output.shape, generator, model_output.device, model_output.dtype)
                noise = noise.to(sample.device).contiguous()
                nptr = _lib.ptr(noise)
        x, e = sample.contiguous(), model_output.contiguous()
        prev = torch.empty_like(x)
        clip = self.config.clip_sample_range if self.config.clip_sample else 0.0
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dsg_ddpm_step(_lib.ptr(x), _lib.ptr(e), nptr, _lib.ptr(prev), x.numel(),
                                                s["sqrt_beta_prod_t"], s["sqrt_alpha_prod_t"], clip, s["coef_x0"],
                                                s["coef_xt"], s["sigma"], _lib.stream_ptr(x.device)))
            if host is not None:
                host.consumed = torch.cuda.Event()
                host.consumed.record(torch.cuda.current_stream(x.device))
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev)

    # ---- config I/O (App. A.5) ----------------------------------------------------------------------
    def save_pretrained(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        cfg = {"_class_name": self._class_name, "_diffusers_version": "0.20.0"}
        cfg.update(self.config.to_dict())
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
            f.write("\n")

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **_unused):
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = json.load(f)
        cfg = {k: v for k, v in cfg.items() if k in cls._defaults}
        return cls(**cfg)

    @classmethod
    def from_config(cls, config):
        cfg = config.to_dict() if hasattr(config, "to_dict") else dict(config)
        return cls(**{k: v for k, v in cfg.items() if k in cls._defaults})


class DDIMScheduler(DDPMScheduler):
    _class_name = "DDIMScheduler"
    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                     prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                     clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                     rescale_betas_zero_snr=False)

    def _check_extra(self, cfg):
        if cfg["rescale_betas_zero_snr"]:
            raise NotImplementedError("DDIMScheduler: rescale_betas_zero_snr is outside the DriveSceneGen path")

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.final_alpha_cumprod = torch.tensor(1.0) if self.config.set_alpha_to_one else self.alphas_cumprod[0]

    def step_scalars(self, t: int, eta: float = 0.0):
        key = (t, self.previous_timestep(t), float(eta))
        hit = self._scalar_cache.get(key)
        if hit is None:
            hit = self._scalar_cache[key] = self._step_scalars(t, eta)
        return hit

    def _step_scalars(self, t: int, eta: float = 0.0):
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        variance = (b_prev / b_t) * (1 - a_t / a_prev)
        std = eta * variance ** 0.5
        dirc = (1 - a_prev - std ** 2) ** 0.5
        return dict(sqrt_beta_prod_t=float(b_t ** 0.5), sqrt_alpha_prod_t=float(a_t ** 0.5),
                    sqrt_alpha_prev=float(a_prev ** 0.5), dir_coef=float(dirc), std=float(std))

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict: bool = True):
        if use_clipped_model_output:
            raise NotImplementedError("DDIMScheduler.step: use_clipped_model_output is outside the benchmarked path")
        if not sample.is_cuda:
            raise RuntimeError("DDIMScheduler.step runs on the MI355X HIP engine only (got a CPU tensor)")
        t = int(timestep)
        s = self.step_scalars(t, eta)
        x, e = sample.contiguous(), model_output.contiguous()
        prev = torch.empty_like(x)
        clip = self.config.clip_sample_range if self.config.clip_sample else 0.0
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().dsg_ddim_step(_lib.ptr(x), _lib.ptr(e), _lib.ptr(prev), x.numel(),
                                                s["sqrt_beta_prod_t"], s["sqrt_alpha_prod_t"], clip,
                                                s["sqrt_alpha_prev"], s["dir_coef"], _lib.stream_ptr(x.device)))
        if eta > 0:
            z = variance_noise if variance_noise is not None else _randn_like_reference(
                model_output.shape, generator, model_output.device, model_output.dtype)
            prev = prev + s["std"] * z.to(prev.device)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev)
