"""drivescenegen_amd -- MI355X (gfx950) engine for DriveSceneGen's denoising hot path.

Drop-in for the objects /root/reference/DriveSceneGen/{scripts/train.py, scripts/generation.py,
pipeline/training_pipeline.py} import from diffusers: same names, arguments, state-dict keys and
checkpoint folder; all arithmetic runs in hand-written HIP kernels behind the C ABI of include/dsg.h.
"""
from .unet import UNet2DModel  # noqa: F401
from .schedulers import DDPMScheduler, DDIMScheduler  # noqa: F401
from .pipelines import DDPMPipeline, DDIMPipeline, ImagePipelineOutput  # noqa: F401
from .optimization import get_cosine_schedule_with_warmup  # noqa: F401
from .train_loop import fit, notebook_launcher, sample_to_pil  # noqa: F401
from .training import AdamW, Accelerator, GradBuckets, clip_grad_norm_, mse_loss  # noqa: F401

__version__ = "0.1.0"
