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
  per batch     noise ~ N(0,1) drawn on the HOST and moved (the reference's `torch.randn(shape).to(device)`), t ~ U{0..T-1}
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


def train_step(accelerator, model, noise_scheduler, optimizer, lr_scheduler, batch):
    """One optimisation step on a clean batch [B, C, H, W] in [-1, 1]; returns the detached loss (a device scalar)."""
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


def fit(config, model, noise_scheduler, optimizer, train_dataloader, lr_scheduler, sample_steps: int = 750,
        on_step=None):
    """Train for ``config.num_epochs`` epochs.  `config` carries the reference's TrainingConfig fields (train.py:13-29):
    mixed_precision, gradient_accumulation_steps, output_dir, num_epochs, save_image_epochs, save_model_epochs,
    eval_batch_size, seed.  Returns the number of optimisation steps taken on this rank."""
    accelerator = Accelerator(mixed_precision=config.mixed_precision,
                              gradient_accumulation_steps=config.gradient_accumulation_steps, log_with="tensorboard",
                              project_dir=os.path.join(config.output_dir, "logs"))
    if accelerator.is_main_process:
        os.makedirs(config.output_dir, exist_ok=True)
        accelerator.init_trackers("train_example")
    model, optimizer, train_dataloader, lr_scheduler = accelerator.prepare(model, optimizer, train_dataloader, lr_scheduler)
    step = 0
    for epoch in range(config.num_epochs):
        for batch in train_dataloader:
            loss = train_step(accelerator, model, noise_scheduler, optimizer, lr_scheduler, batch)
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
