"""Training-mode forward (autograd) of UNet2DModel on the HIP engine -- see training.py."""


def unet_forward_train(model, sample, timestep):
    raise NotImplementedError(
        "drivescenegen_amd: the differentiable (training) forward is not built yet; wrap inference in "
        "torch.no_grad() or call model.requires_grad_(False)")
