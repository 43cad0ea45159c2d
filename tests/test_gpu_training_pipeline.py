"""End-to-end: the reference's train loop shape (training_pipeline.py:46-107) on the engine -- dataset ->
add_noise -> U-Net -> mse -> backward -> clip -> AdamW -> cosine LR -> evaluate (seeded sampling, PNG) ->
save_pretrained -> generation-style reload."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import synth  # noqa: E402
from drivescenegen_amd.dataset import Image_Dataset  # noqa: E402
from drivescenegen_amd.training_pipeline import TrainingPipeline  # noqa: E402
from tests.common import CFG1, synth_weights  # noqa: E402


def _config(tmp_path, n_img=8):
    from PIL import Image
    data = tmp_path / "data"
    data.mkdir()
    x = synth.synth_scene_rasters(n_img, 3, 64, 64, 3)
    for i in range(n_img):
        Image.fromarray(((x[i].transpose(1, 2, 0) * 0.5 + 0.5) * 255).round().astype(np.uint8)).save(data / f"{i}.png")
    return SimpleNamespace(patterns_size_height=64, patterns_size_width=64, train_batch_size=4, eval_batch_size=1,
                           num_epochs=2, gradient_accumulation_steps=1, learning_rate=2e-4, lr_warmup_steps=2,
                           save_image_epochs=1, save_model_epochs=1, mixed_precision="fp16",
                           output_dir=str(tmp_path / "out"), dataset_name=str(data / "*"), overwrite_output_dir=True,
                           seed=14555, num_inference_steps=5)


def test_train_loop_end_to_end(tmp_path):
    config = _config(tmp_path)
    dataset = Image_Dataset(config)
    assert len(dataset) == 8 and dataset[0].shape == (3, 64, 64)
    loader = torch.utils.data.DataLoader(dataset, batch_size=config.train_batch_size, shuffle=True)
    torch.manual_seed(0)
    model = synth_weights(d.UNet2DModel(**CFG1))
    noise_scheduler = d.DDPMScheduler()
    optimizer = d.AdamW(model.parameters(), lr=config.learning_rate)
    lr_scheduler = d.get_cosine_schedule_with_warmup(optimizer=optimizer, num_warmup_steps=config.lr_warmup_steps,
                                                     num_training_steps=len(loader) * config.num_epochs)
    hist = TrainingPipeline(config).train_loop(config, model, noise_scheduler, optimizer, loader, lr_scheduler)
    assert len(hist) == 4 and all(np.isfinite(h["loss"]) for h in hist)
    assert hist[0]["lr"] == pytest.approx(config.learning_rate / 2) and hist[1]["lr"] == pytest.approx(
        config.learning_rate)
    out = config.output_dir
    for rel in ("model_index.json", "unet/config.json", "unet/diffusion_pytorch_model.bin",
                "scheduler/scheduler_config.json", "samples/000.png", "samples/001.png", "logs/train_example.jsonl"):
        assert os.path.exists(os.path.join(out, rel)), rel
    from PIL import Image
    im = Image.open(os.path.join(out, "samples/000.png"))
    assert im.size == (64, 64) and im.mode == "RGB"
    # generation.py: reload and sample
    ddpm = d.DDPMPipeline.from_pretrained(out, variant="fp16").to("cuda")
    ddpm.unet.requires_grad_(False)
    imgs = ddpm(batch_size=2, num_inference_steps=3).images
    assert len(imgs) == 2 and imgs[0].size == (64, 64)


def test_loss_goes_down_on_fixed_batch():
    """A few AdamW steps on one fixed (batch, noise, t) reduce the loss: the whole fwd/bwd/update chain is wired
    with the right signs."""
    net = synth_weights(d.UNet2DModel(**CFG1)).to("cuda").train()
    opt = d.AdamW(net.parameters(), lr=1e-3)
    sch = d.DDPMScheduler()
    x0 = torch.from_numpy(synth.synth_scene_rasters(4, 3, 64, 64, 9)).cuda()
    noise = torch.from_numpy(synth.normal(10, (4, 3, 64, 64))).cuda()
    t = torch.tensor([50, 300, 600, 900], device="cuda")
    noisy = sch.add_noise(x0, noise, t)
    losses = []
    for _ in range(8):
        loss = d.mse_loss(net(noisy, t, return_dict=False)[0], noise)
        loss.backward()
        d.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.7 * losses[0], losses
