"""End-to-end training driver on the engine's objects: epochs of DDPM epsilon-prediction steps, a seeded sample PNG and a
diffusers-layout checkpoint at the configured epoch intervals.

What it stands in for: the reference's driver, /root/reference/DriveSceneGen/pipeline/training_pipeline.py
(``TrainingPipeline.train_loop`` :46-107, ``.evaluate`` :16-43), and the launcher call at
/root/reference/DriveSceneGen/scripts/train.py:121-122 (``accelerate.notebook_launcher``).  After the import swap of
INTEGRATION.md section 1 the reference's own file runs on these objects as it is, EXCEPT for two imports the swap does not
reach: ``from torchvision import transforms`` (training_pipeline.py:2, used only by ``evaluate`` to turn the sample into a
PIL image) and ``from accelerate import notebook_launcher`` (train.py:4).  Both have counterparts here
(``sample_to_pil``, ``notebook_launcher``), and ``fit`` is the whole loop for users who do not carry the reference's file.

Written from the behaviour listed in SURVEY.md sections 3.1-3.2 / App. A.4-A.6, not from the reference's text:
  per batch     noise ~ N(0,1) drawn on the HOST and moved (the reference's `torch.randn(shape).to(device)`; `NoiseAhead`
                draws the NEXT step's tensor on a worker thread while the GPU runs this one -- same values), t ~ U{0..T-1}
                drawn on the device, x_t = add_noise(x0, noise, t); inside accelerator.accumulate: eps = model(x_t, t),
                loss = mse(eps, noise), backward, clip to 1.0, optimizer / LR-scheduler step, zero_grad; log loss, lr, step
  per epoch     on the main process, when (epoch+1) % save_image_epochs == 0 or it is the last: one 750-step DDPM sample
                from torch.manual_seed(config.seed), image 0 scaled by 255 and TRUNCATED to uint8 -> samples/NNN.png;
                when (epoch+1) % save_model_epochs == 0 or last: pipeline.save_pretrained(output_dir)
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .pipelines import DDPMPipeline
from .training import Accelerator, mse_loss


def sample_to_pil(image_hwc):
    """[H, W, C] float in [0, 1] -> PIL image the way the reference's evaluate does it (x * 255, numpy ``astype(uint8)``
    = truncation toward zero, then ToPILImage): the torchvision-free counterpart of training_pipeline.py:18-24."""
    from PIL import Image
    a = (torch.as_tensor(image_hwc) * 255.0).numpy().astype(np.uint8)
    return Image.fromarray(a[:, :, 0], mode="L") if a.shape[2] == 1 else Image.fromarray(a)


def write_sample(config, pipeline, steps: int = 750):
    """One seeded sample -> ``<output_dir>/samples/NNN.png`` (NNN = files already there); returns the path."""
    out = pipeline(num_inference_steps=steps, batch_size=config.eval_batch_size, generator=torch.manual_seed(config.seed),
                   output_type="np.array", return_dict=False)
    first = torch.from_numpy(np.asarray(out))[0, 0]   # the 1-tuple of [B, H, W, C] -> image 0 of the batch
    folder = os.path.join(config.output_dir, "samples")
    os.makedirs(folder, exist_ok=True)
    index = sum(os.path.isfile(os.path.join(folder, f)) for f in os.listdir(folder))
    path = os.path.join(folder, f"{index:03d}.png")
    sample_to_pil(first).save(path)
    return path


class NoiseAhead:
    """The host-side noise draw of the training step (training_pipeline.py:72: ``torch.randn(batch.shape).to(device)``,
    2.75 M normals = ~8 ms of a 34-ms step at the reference's batch 14) moved off the critical path: a worker thread
    draws step k+1's tensor into pinned memory while the GPU runs step k.

    The values are the serial loop's, bit for bit: the same global CPU generator, the same ``torch.randn(shape)`` calls
    in the same order.  ``schedule(shape)`` is only ever called for a batch that has already been fetched, from the
    thread that would otherwise draw, while no other consumer of the global generator runs -- ``fit`` fetches batch k+1
    (whose sampler seed, if any, was drawn when the epoch's iterator was made) BEFORE it schedules, and joins the worker
    before anything else may draw (``take``).  One draw in flight at most."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled)
        self._thread = None
        self._shape = None
        self._out = None
        self._snap = None
        self._side = None      # the worker's copy stream: the 268-MB H2D of configs[4]'s tensor does not sit in the training stream

    def schedule(self, shape, device=None):
        if not self.enabled or self._thread is not None:
            return
        import threading
        self._shape = tuple(shape)
        self._snap = torch.get_rng_state()    # the global CPU generator before the draw (cancel() goes back to it)
        dev = torch.device(device) if device is not None else None
        if dev is not None and dev.type == "cuda" and self._side is None:
            self._side = torch.cuda.Stream(dev)

        def draw():
            try:
                host = torch.randn(self._shape, pin_memory=torch.cuda.is_available())
                if dev is not None and dev.type == "cuda":   # copy ahead too, on the worker's own stream, an event behind it
                    with torch.cuda.stream(self._side):
                        on_dev = host.to(dev, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(self._side)
                    self._out = (host, on_dev, ev)
                else:
                    self._out = (host, None, None)
            except BaseException as e:  # surfaced by take() on the training thread
                self._out = e
        self._thread = threading.Thread(target=draw, daemon=True)
        self._thread.start()

    def cancel(self):
        """The consumer stops before using the scheduled draw (an exception or `break` in its loop, a max-steps cut): join the
        worker and put the global CPU generator back where the serial loop would be -- the draw never happened."""
        if self._thread is not None:
            self._thread.join()
            self._thread, self._out = None, None
            torch.set_rng_state(self._snap)

    def take(self, shape, device):
        """The noise of the current step on `device`: the scheduled draw when its shape matches, else a draw made now."""
        shape = tuple(shape)
        if self._thread is not None:
            self._thread.join()
            out, self._thread, self._out = self._out, None, None
            if isinstance(out, BaseException):
                raise out
            if self._shape == shape:
                host, on_dev, ev = out
                if on_dev is not None and on_dev.device == torch.device(device):
                    cur = torch.cuda.current_stream(on_dev.device)
                    cur.wait_event(ev)
                    on_dev.record_stream(cur)
                    return on_dev
                return host.to(device, non_blocking=True)
            # (cannot happen in `fit`: it schedules the very batch it takes next): undo the draw, then say so
            torch.set_rng_state(self._snap)
            raise RuntimeError(f"NoiseAhead: scheduled {self._shape}, asked for {shape}")
        return torch.randn(shape).to(device)


class DeviceNoise:
    """Opt-in replacement of the host draw: the step's noise comes from the library's counter-based generator
    (Philox4x32-10 + Box-Muller, ``dsg_add_noise_philox``) in the same pass that forms x_t -- no serial CPU draw
    (500 ms for configs[4]'s [128, 8, 256, 256] on one host thread against a 183-ms GPU step), no 268-MB H2D copy, one tensor
    read less.  NOT the reference's values (those are the global CPU generator's): the default stays the host draw.
    The stream of step k on rank r is (seed, offset = r << 40 | k): reproducible, rank-disjoint, resumable from `step`."""

    def __init__(self, seed: int = 0, rank: int = 0, step: int = 0):
        self.seed, self.rank, self.step = int(seed), int(rank), int(step)

    def next_offset(self) -> int:
        off = (self.rank << 40) | self.step
        self.step += 1
        return off


def train_step(accelerator, model, noise_scheduler, optimizer, lr_scheduler, batch, noise=None):
    """One optimisation step on a clean batch [B, C, H, W] in [-1, 1]; returns the detached loss (a device scalar).
    `noise`: the step's N(0,1) tensor on the batch's device when the caller drew it ahead (``NoiseAhead``); a
    ``DeviceNoise`` for the opt-in library generator; drawn here, on the host, otherwise -- the reference's own order."""
    t = None
    if isinstance(noise, DeviceNoise):
        t = torch.randint(0, noise_scheduler.num_train_timesteps, (batch.shape[0],), device=batch.device).long()
        noisy, noise = noise_scheduler.add_noise_device(batch, t, noise.seed, noise.next_offset())
    else:
        if noise is None:
            noise = torch.randn(batch.shape).to(batch.device)
        t = torch.randint(0, noise_scheduler.num_train_timesteps, (batch.shape[0],), device=batch.device).long()
        noisy = noise_scheduler.add_noise(batch, noise, t).to(torch.float)
    with accelerator.accumulate(model):
        loss = mse_loss(model(noisy, t, return_dict=False)[0], noise)
        accelerator.backward(loss)
        accelerator.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        lr_scheduler.step()
        optimizer.zero_grad()
    return loss.detach()


def batches_with_noise(batches, overlap_noise: bool = True):
    """One pass over `batches` as (batch, noise) pairs, the noise of batch k+1 being drawn on the worker thread while the
    consumer works on batch k.  Consumers of the global CPU generator keep the serial loop's order -- fetch k, draw k,
    fetch k+1, draw k+1, ... -- and never overlap: the worker is joined before the next fetch."""
    ahead = NoiseAhead(overlap_noise)
    it = iter(batches)
    batch = next(it, None)
    try:
        while batch is not None:
            noise = ahead.take(batch.shape, batch.device)
            nxt = next(it, None)
            if nxt is not None:
                ahead.schedule(nxt.shape, nxt.device)
            yield batch, noise
            batch = nxt
    finally:   # a consumer that stops early (exception, break, max-steps cut) leaves the generator as the serial loop would
        ahead.cancel()


def train_steps(accelerator, model, noise_scheduler, optimizer, lr_scheduler, batches, overlap_noise: bool = True,
                noise="host"):
    """Generator over one pass of `batches` (an epoch's loader): yields each step's detached loss.  Batch k+1 is fetched and
    its noise draw handed to the worker thread before step k's kernels are queued, so the host draw overlaps the GPU.
    ``noise="device"`` (or a ``DeviceNoise``): the library's counter-based generator instead of the host draw (opt-in)."""
    if noise != "host":
        gen = noise if isinstance(noise, DeviceNoise) else DeviceNoise(rank=getattr(accelerator, "process_index", 0))
        if noise != "device" and not isinstance(noise, DeviceNoise):
            raise ValueError(f"train_steps: noise must be 'host', 'device' or a DeviceNoise (got {noise!r})")
        for batch in batches:
            yield train_step(accelerator, model, noise_scheduler, optimizer, lr_scheduler, batch, noise=gen)
        return
    for batch, nz in batches_with_noise(batches, overlap_noise):
        yield train_step(accelerator, model, noise_scheduler, optimizer, lr_scheduler, batch, noise=nz)


def fit(config, model, noise_scheduler, optimizer, train_dataloader, lr_scheduler, sample_steps: int = 750,
        on_step=None, overlap_noise: bool = True, noise="host"):
    """Train for ``config.num_epochs`` epochs.  `config` carries the reference's TrainingConfig fields (train.py:13-29):
    mixed_precision, gradient_accumulation_steps, output_dir, num_epochs, save_image_epochs, save_model_epochs,
    eval_batch_size, seed.  Returns the number of optimisation steps taken on this rank.  `overlap_noise=False` draws each
    step's noise on the critical path like the reference does (same values either way).  ``noise="device"``: the library's
    counter-based generator (seeded from ``config.seed``; one stream per rank and step) instead of the host draw -- opt-in,
    not the reference's values."""
    accelerator = Accelerator(mixed_precision=config.mixed_precision,
                              gradient_accumulation_steps=config.gradient_accumulation_steps, log_with="tensorboard",
                              project_dir=os.path.join(config.output_dir, "logs"))
    if accelerator.is_main_process:
        os.makedirs(config.output_dir, exist_ok=True)
        accelerator.init_trackers("train_example")
    model, optimizer, train_dataloader, lr_scheduler = accelerator.prepare(model, optimizer, train_dataloader, lr_scheduler)
    step = 0
    if noise == "device":
        noise = DeviceNoise(seed=getattr(config, "seed", 0), rank=accelerator.process_index)
    for epoch in range(config.num_epochs):
        for loss in train_steps(accelerator, model, noise_scheduler, optimizer, lr_scheduler, train_dataloader, overlap_noise,
                                noise=noise):
            record = {"loss": loss.item(), "lr": lr_scheduler.get_last_lr()[0], "step": step}
            accelerator.log(record, step=step)
            if on_step is not None:
                on_step(epoch, record)
            step += 1
        last = epoch == config.num_epochs - 1
        if accelerator.is_main_process:
            pipeline = DDPMPipeline(unet=accelerator.unwrap_model(model), scheduler=noise_scheduler)
            if last or (epoch + 1) % config.save_image_epochs == 0:
                write_sample(config, pipeline, sample_steps)
            if last or (epoch + 1) % config.save_model_epochs == 0:
                pipeline.save_pretrained(config.output_dir)
    accelerator.end_training()
    return step


def notebook_launcher(function, args=(), num_processes: int = 1, **_ignored):
    """``accelerate.notebook_launcher`` for the way train.py:122 uses it (``num_processes=1``): call the function here.
    More than one process per node is what ``python -m torch.distributed.run --nproc-per-node N`` is for -- one process
    per GPU, RANK / LOCAL_RANK / WORLD_SIZE from the environment, which is what ``Accelerator`` reads."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if num_processes not in (1, world):
        raise RuntimeError(f"notebook_launcher(num_processes={num_processes}): start {num_processes} processes with "
                           "`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 <script>` instead")
    return function(*args)
