"""``TrainingPipeline`` -- DriveSceneGen's train loop and per-epoch sampler, on the MI355X engine.

Mirror of /root/reference/DriveSceneGen/pipeline/training_pipeline.py:11-107: same class, method names,
arguments, step order and side effects (samples/NNN.png, save_pretrained every epoch).  Differences, all
forced by the image (no torchvision / tensorboard / tqdm dependency on the hot path):
 - ``accelerate.Accelerator`` -> drivescenegen_amd.Accelerator (same calls), logs go to JSONL;
 - ``F.mse_loss`` -> drivescenegen_amd.mse_loss (HIP kernel);
 - the PIL conversion of ``evaluate`` is written out (x*255 -> uint8 truncation -> PIL), as torchvision's
   ToPILImage does for a uint8 HWC array.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .pipelines import DDPMPipeline
from .training import Accelerator, mse_loss


class TrainingPipeline:
    def __init__(self, config):
        self.config = config

    # ---- evaluate (training_pipeline.py:16-43) ----
    def evaluate(self, config, epoch, pipeline):
        polylines_patterns = pipeline(
            num_inference_steps=getattr(config, "num_inference_steps", 750),
            batch_size=config.eval_batch_size,
            generator=torch.manual_seed(config.seed),
            output_type="np.array",
            return_dict=False,
        )
        test_dir = os.path.join(config.output_dir, "samples")
        os.makedirs(test_dir, exist_ok=True)
        file_count = len([f for f in os.listdir(test_dir) if os.path.isfile(os.path.join(test_dir, f))])
        polylines_patterns = torch.tensor(np.asarray(polylines_patterns))
        polylines_patterns = polylines_patterns[0, 0, :, :, :]
        arr = (polylines_patterns * 255.).numpy().astype(np.uint8)  # truncation, as the reference
        from PIL import Image
        img = Image.fromarray(arr[:, :, 0], mode="L") if arr.shape[-1] == 1 else Image.fromarray(arr)
        img.save(f"{test_dir}/" + f"{file_count:03d}" + ".png")

    # ---- train loop (training_pipeline.py:46-107) ----
    def train_loop(self, config, model, noise_scheduler, optimizer, train_dataloader, lr_scheduler,
                   max_steps=None, progress=None):
        accelerator = Accelerator(
            mixed_precision=config.mixed_precision,
            gradient_accumulation_steps=config.gradient_accumulation_steps,
            log_with="jsonl",
            project_dir=os.path.join(config.output_dir, "logs"),
        )
        if accelerator.is_main_process:
            os.makedirs(config.output_dir, exist_ok=True)
            accelerator.init_trackers("train_example")

        model, optimizer, train_dataloader, lr_scheduler = accelerator.prepare(
            model, optimizer, train_dataloader, lr_scheduler)

        global_step = 0
        history = []
        for epoch in range(config.num_epochs):
            for step, batch in enumerate(train_dataloader):
                # noise on the CPU global RNG, then moved (training_pipeline.py:72)
                noise = torch.randn(batch.shape).to(batch.device)
                bs = batch.shape[0]
                timesteps = torch.randint(0, noise_scheduler.num_train_timesteps, (bs,), device=batch.device).long()
                noisy_pattern = noise_scheduler.add_noise(batch, noise, timesteps).to(torch.float)

                with accelerator.accumulate(model):
                    noise_pred = model(noisy_pattern, timesteps, return_dict=False)[0]
                    loss = mse_loss(noise_pred, noise)
                    accelerator.backward(loss)
                    accelerator.clip_grad_norm_(model.parameters(), 1.0)
                    optimizer.step()
                    lr_scheduler.step()
                    optimizer.zero_grad()

                logs = {"loss": loss.detach().item(), "lr": lr_scheduler.get_last_lr()[0], "step": global_step}
                if progress is not None:
                    progress(logs)
                accelerator.log(logs, step=global_step)
                history.append(logs)
                global_step += 1
                if max_steps is not None and global_step >= max_steps:
                    break

            if accelerator.is_main_process:
                pipeline = DDPMPipeline(unet=accelerator.unwrap_model(model), scheduler=noise_scheduler)
                if (epoch + 1) % config.save_image_epochs == 0 or epoch == config.num_epochs - 1:
                    self.evaluate(config, epoch, pipeline)
                if (epoch + 1) % config.save_model_epochs == 0 or epoch == config.num_epochs - 1:
                    pipeline.save_pretrained(config.output_dir)
            if max_steps is not None and global_step >= max_steps:
                break
        accelerator.end_training()
        return history
