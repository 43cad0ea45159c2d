"""``get_cosine_schedule_with_warmup`` as imported at /root/reference/DriveSceneGen/scripts/train.py:3,67-71
(diffusers.optimization).  Host-side scalar logic only."""
from __future__ import annotations

import math

from torch.optim.lr_scheduler import LambdaLR


def cosine_with_warmup_lambda(step: int, num_warmup_steps: int, num_training_steps: int,
                              num_cycles: float = 0.5) -> float:
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int,
                                    num_cycles: float = 0.5, last_epoch: int = -1):
    def lr_lambda(current_step):
        return cosine_with_warmup_lambda(current_step, num_warmup_steps, num_training_steps, num_cycles)

    return LambdaLR(optimizer, lr_lambda, last_epoch)
