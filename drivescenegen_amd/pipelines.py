"""``DDPMPipeline`` / ``DDIMPipeline`` -- the sampler objects DriveSceneGen's scripts call.

Reference call sites:
 - /root/reference/DriveSceneGen/pipeline/training_pipeline.py:101 ``DDPMPipeline(unet=..., scheduler=...)``,
   :26-32 ``pipeline(num_inference_steps=750, batch_size, generator=torch.manual_seed(seed),
   output_type="np.array", return_dict=False)``, :107 ``pipeline.save_pretrained(output_dir)``;
 - /root/reference/DriveSceneGen/scripts/generation.py:7 ``DDPMPipeline.from_pretrained(path, variant="fp16")
   .to('cuda')`` and :14-20 ``ddpm(batch_size=5, num_inference_steps=750).images``.
Semantics: SURVEY.md App. A.4 (noise stream order, post-processing) and A.5 (folder layout).

The denoising loop stays host-driven like the reference's, but each iteration is two asynchronous C-ABI
calls on the current HIP stream (dsg_unet_forward + dsg_ddpm_step / dsg_ddim_step); batched inference
shards samples over ranks with no collective (``shard=(rank, world)``).
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from .schedulers import DDIMScheduler, DDPMScheduler, _randn_like_reference
from .unet import UNet2DModel


class ImagePipelineOutput(SimpleNamespace):
    pass


def numpy_to_pil(images: np.ndarray):
    """diffusers ``numpy_to_pil``: (x*255).round().astype(uint8); C=1 -> mode L."""
    from PIL import Image
    if images.ndim == 3:
        images = images[None, ...]
    images = (images * 255).round().astype("uint8")
    if images.shape[-1] == 1:
        return [Image.fromarray(im.squeeze(), mode="L") for im in images]
    if images.shape[-1] not in (3, 4):
        raise ValueError(f"PIL output supports 1, 3 or 4 channels (got {images.shape[-1]}); "
                         "use output_type='np.array'")
    return [Image.fromarray(im) for im in images]


class _PipelineBase:
    _class_name = "DDPMPipeline"
    _scheduler_cls = DDPMScheduler

    def __init__(self, unet, scheduler):
        self.unet = unet
        self.scheduler = scheduler

    @property
    def device(self):
        return self.unet.device

    def to(self, device):
        self.unet.to(device)
        return self

    # ---- App. A.5 folder layout -------------------------------------------------------------------
    def save_pretrained(self, save_directory, safe_serialization: bool = False, variant=None):
        os.makedirs(save_directory, exist_ok=True)
        index = {"_class_name": self._class_name, "_diffusers_version": "0.20.0",
                 "scheduler": ["diffusers", type(self.scheduler)._class_name],
                 "unet": ["diffusers", "UNet2DModel"]}
        with open(os.path.join(save_directory, "model_index.json"), "w") as f:
            json.dump(index, f, indent=2, sort_keys=True)
            f.write("\n")
        self.unet.save_pretrained(os.path.join(save_directory, "unet"), safe_serialization=safe_serialization,
                                  variant=variant)
        self.scheduler.save_pretrained(os.path.join(save_directory, "scheduler"))

    @classmethod
    def from_pretrained(cls, path, variant=None, torch_dtype=None, **_unused):
        with open(os.path.join(path, "model_index.json")) as f:
            index = json.load(f)
        sched_name = index.get("scheduler", ["diffusers", "DDPMScheduler"])[1]
        sched_cls = {"DDPMScheduler": DDPMScheduler, "DDIMScheduler": DDIMScheduler}.get(sched_name)
        if sched_cls is None:
            raise NotImplementedError(f"scheduler class {sched_name!r} not supported")
        if cls._scheduler_cls is not sched_cls:
            # diffusers lets a pipeline adopt a compatible scheduler config
            scheduler = cls._scheduler_cls.from_config(
                {k: v for k, v in json.load(open(os.path.join(path, "scheduler", "scheduler_config.json"))).items()})
        else:
            scheduler = sched_cls.from_pretrained(path, subfolder="scheduler")
        unet = UNet2DModel.from_pretrained(path, subfolder="unet", variant=variant, torch_dtype=torch_dtype)
        return cls(unet=unet, scheduler=scheduler)

    # ---- shared loop --------------------------------------------------------------------------------
    def _initial_noise(self, batch_size, generator, shard):
        c = self.unet.config
        ss = c.sample_size
        shape = (batch_size, c.in_channels, ss, ss) if isinstance(ss, int) else (batch_size, c.in_channels, *ss)
        image = _randn_like_reference(shape, generator, self.device, torch.float32)
        return self._shard(image, shard)

    @staticmethod
    def _shard(x, shard):
        if shard is None:
            return x
        rank, world = shard
        b = x.shape[0]
        if b % world != 0:
            raise ValueError(f"batch {b} not divisible by world size {world}")
        per = b // world
        return x[rank * per:(rank + 1) * per].contiguous()

    def _finish(self, image, output_type, return_dict):
        b, c, h, w = image.shape
        out = torch.empty((b, h, w, c), dtype=torch.float32, device=image.device)
        with torch.cuda.device(image.device):
            _lib.check(_lib.load().dsg_postprocess(_lib.ptr(image.contiguous()), _lib.ptr(out), b, c, h * w, 0,
                                                  _lib.stream_ptr(image.device)))
        arr = out.cpu().numpy()
        if output_type == "pil":
            arr = numpy_to_pil(arr)
        if not return_dict:
            return (arr,)
        return ImagePipelineOutput(images=arr)


class _NoiseStream:
    """The per-step noise of ``DDPMPipeline.__call__`` in the reference's order of draws (App. A.4: x_T, then one full-batch
    tensor per step with t > 0, all from ONE generator).  A CPU generator (training_pipeline.py:29 passes
    ``torch.manual_seed(seed)``) samples on the host: tensor k+1 is drawn by a worker thread into one of THREE pinned buffers while
    the caller enqueues step k's kernels, and the scheduler's step kernel reads that buffer in place, over PCIe
    (``schedulers.HostNoise``: the noise is read exactly once, 12 bytes per pixel of a batch-1 sample) -- same generator, same
    shapes, same order, so the same values as ``randn_tensor(...).to(device)`` bit for bit, with neither the draw, nor a
    pageable copy, nor a second stream's hand-over on the critical path.  Measured on the evaluate call (750 steps, batch 1;
    tools/whole_call_probe.py): device-RNG loop 2.15 ms per step; this 2.18; a pinned copy on the same stream 2.20; a copy on
    a side stream with event hand-overs 2.27.  `count` = the draws the call will make: the worker never draws past it, and
    ``close()`` (always called) joins it and, should the loop have ended early, puts the generator back where the serial
    loop would have left it.  With no generator, or a device generator, a draw is torch's device RNG kernel on the current
    stream, as in the reference."""

    RING = 3

    def __init__(self, shape, generator, device, rows=None, count=1):
        self.shape, self.gen, self.dev, self.rows, self.count = tuple(shape), generator, torch.device(device), rows, int(count)
        self.host = generator is not None and generator.device.type == "cpu"
        self.k = 0
        self._pending = None          # (thread, index, generator state before the draw, [exception])
        if self.host:
            # three buffers: the draw of tensor k waits for the kernel that read tensor k - 3, so the host may run two whole
            # steps ahead of the GPU (with two it stalled on the step before last: never more than one forward queued)
            self.pinned = [torch.empty(self.shape, dtype=torch.float32).pin_memory() for _ in range(self.RING)]
            self.lent = [None] * self.RING  # the HostNoise last handed out on pinned[i] (its `consumed` event guards the reuse)

    def _start(self, k):
        """hand tensor k's draw to a worker thread (one in flight at most; nobody else touches the generator meanwhile)"""
        import threading
        i = k % self.RING
        if self.lent[i] is not None:
            self.lent[i].wait_consumed()      # the step kernel that read pinned[i] three draws ago has run
            self.lent[i] = None
        box = []

        def work():
            try:
                torch.randn(self.shape, generator=self.gen, out=self.pinned[i])
            except BaseException as e:   # surfaced by draw()
                box.append(e)
        snap = self.gen.get_state()
        th = threading.Thread(target=work, daemon=True)
        th.start()
        self._pending = (th, k, snap, box)

    def close(self):
        if self._pending is not None:     # a tensor was drawn ahead and never asked for: undo the draw
            th, _, snap, _ = self._pending
            th.join()
            self.gen.set_state(snap)
            self._pending = None
        if self.host:
            for h in self.lent:           # the pinned buffers must outlive the kernels that read them
                if h is not None:
                    h.wait_consumed()
            self.lent = [None] * self.RING

    def draw(self):
        """The next tensor of the stream (this shard's rows of it): a device tensor, or -- host generator, after x_T -- a
        ``HostNoise`` over a pinned buffer, to be handed to ``scheduler.step(..., variance_noise=...)`` before the draw after
        next."""
        from .schedulers import HostNoise
        rows = self.rows
        k, self.k = self.k, self.k + 1
        if not self.host:
            z = torch.randn(self.shape, generator=self.gen, device=self.dev, dtype=torch.float32)
            return z if rows is None else z[rows].contiguous()
        if k == 0:   # x_T is the U-Net's first input: a device tensor, the reference's plain path
            z = torch.randn(self.shape, generator=self.gen, dtype=torch.float32)
            out = (z if rows is None else z[rows]).to(self.dev)
            if self.count > 1:
                self._start(1)
            return out
        i = k % self.RING
        if self._pending is None:             # (more draws than announced: drawn here, in order)
            self._start(k)
        th, kk, _, box = self._pending
        th.join()
        self._pending = None
        if box:
            raise box[0]
        assert kk == k
        out = HostNoise(self.pinned[i] if rows is None else self.pinned[i][rows])
        self.lent[i] = out
        if k + 1 < self.count:
            self._start(k + 1)                # drawn while the caller enqueues the rest of this step and the next forward
        return out


def _rows_of(shard, batch):
    if shard is None:
        return None
    rank, world = shard
    if batch % world != 0:
        raise ValueError(f"batch {batch} not divisible by world size {world}")
    per = batch // world
    return slice(rank * per, (rank + 1) * per)


def _device_timestep_rows(timesteps, batch, device):
    """[steps, batch] int64 on the device, uploaded ONCE per call: row i is the timestep tensor of step i (the loop then
    hands the U-Net a device view instead of a host scalar -- no H2D copy and no expand kernel per step)."""
    return timesteps.to(device=device, dtype=torch.long)[:, None].expand(-1, batch).contiguous()


class DDPMPipeline(_PipelineBase):
    _class_name = "DDPMPipeline"
    _scheduler_cls = DDPMScheduler

    @torch.no_grad()
    def __call__(self, batch_size: int = 1, generator=None, num_inference_steps: int = 1000, output_type="pil",
                 return_dict: bool = True, shard=None):
        if self.device.type != "cuda":
            raise RuntimeError("DDPMPipeline runs on the MI355X HIP engine only: call .to('cuda') first")
        c = self.unet.config
        ss = c.sample_size
        full = (batch_size, c.in_channels, ss, ss) if isinstance(ss, int) else (batch_size, c.in_channels, *ss)
        rows = _rows_of(shard, batch_size)
        with torch.cuda.device(self.device):
            # reference semantics: one generator stream for the whole batch (App. A.4); a shard draws the full-batch
            # tensors and keeps its rows so that N-GPU output == 1-GPU output
            self.scheduler.set_timesteps(num_inference_steps)
            ts = [int(t) for t in self.scheduler.timesteps]
            stream = _NoiseStream(full, generator, self.device, rows, count=1 + sum(1 for t in ts if t > 0))
            try:
                image = stream.draw()
                tdev = _device_timestep_rows(self.scheduler.timesteps, image.shape[0], self.device)
                for i, t in enumerate(ts):
                    eps = self.unet(image, tdev[i]).sample
                    noise = stream.draw() if t > 0 else None     # (drawn while the previous step was enqueued; PCIe under the forward)
                    image = self.scheduler.step(eps, t, image, variance_noise=noise).prev_sample
            finally:
                stream.close()
        return self._finish(image, output_type, return_dict)


class DDIMPipeline(_PipelineBase):
    _class_name = "DDIMPipeline"
    _scheduler_cls = DDIMScheduler

    @torch.no_grad()
    def __call__(self, batch_size: int = 1, generator=None, eta: float = 0.0, num_inference_steps: int = 50,
                 use_clipped_model_output=None, output_type="pil", return_dict: bool = True, shard=None):
        if self.device.type != "cuda":
            raise RuntimeError("DDIMPipeline runs on the MI355X HIP engine only: call .to('cuda') first")
        image = self._initial_noise(batch_size, generator, shard)
        self.scheduler.set_timesteps(num_inference_steps)
        ts = [int(t) for t in self.scheduler.timesteps]
        tdev = _device_timestep_rows(self.scheduler.timesteps, image.shape[0], self.device)
        for i, t in enumerate(ts):
            eps = self.unet(image, tdev[i]).sample
            image = self.scheduler.step(eps, t, image, eta=eta, generator=generator).prev_sample
        return self._finish(image, output_type, return_dict)
