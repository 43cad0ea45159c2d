"""``Image_Dataset`` -- the feeder of DriveSceneGen's training loop, without torchvision.

Reference: /root/reference/DriveSceneGen/utils/datasets/dataset.py:15-50 (used at scripts/train.py:34-35):
glob ``config.dataset_name``; ``.pkl`` -> ``torch.load(f)['fig_tensor']`` HWC -> CHW (non-dict pickles fall
through to the next index), anything else -> ``ToTensor(Image.open(f))`` ([0,1], CHW); then
``Resize((H, W), antialias=False)`` (bilinear, align_corners=False) and ``Normalize([0.5], [0.5])``.
Host-side I/O and a per-item bilinear resize; the GPU input pipeline is SURVEY row f1.
"""
from __future__ import annotations

import glob
import os

import numpy as np
import torch
import torch.nn.functional as F
import torch.utils.data as torch_data


def to_tensor(img) -> torch.Tensor:
    """torchvision ``ToTensor`` for PIL images: uint8 HWC -> float CHW in [0,1] (other modes: as numpy)."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)
    return t.float().div(255) if t.dtype == torch.uint8 else t.float()


def resize_bilinear(x: torch.Tensor, size) -> torch.Tensor:
    """torchvision ``Resize(size, antialias=False)`` on a CHW float tensor."""
    return F.interpolate(x[None], size=tuple(size), mode="bilinear", align_corners=False, antialias=False)[0]


class Image_Dataset(torch_data.Dataset):
    def __init__(self, config):
        self.data_list = glob.glob(config.dataset_name)
        self.config = config
        self.size = (config.patterns_size_height, config.patterns_size_width)

    def __len__(self):
        return len(self.data_list)

    def remove_sample(self, index):
        del self.data_list[index]

    def normalize(self, sample):
        return (resize_bilinear(sample, self.size) - 0.5) / 0.5

    def __getitem__(self, index):
        # a .pkl that does not hold a dict is skipped for the next file, as the reference does (utils/datasets/dataset.py:37-39)
        # -- bounded by one round of the list; pickles are read with the restricted unpickler unless the config says
        # trust_pickles=True (imageops.load_sample_pickle: the same policy as GpuImageLoader)
        from .imageops import load_sample_pickle
        n = len(self.data_list)
        for k in range(n):
            file = self.data_list[(index + k) % n]
            if os.path.splitext(file)[1].lower() == ".pkl":
                data_dict = load_sample_pickle(file, bool(getattr(self.config, "trust_pickles", False)))
                if not isinstance(data_dict, dict):
                    continue
                sample = data_dict["fig_tensor"][:, :, :].permute(2, 0, 1).float()
            else:
                from PIL import Image
                with open(file, "rb") as f:
                    sample = to_tensor(Image.open(f))
            return self.normalize(sample)
        raise IndexError(f"Image_Dataset: no usable sample among {n} files (every .pkl holds a non-dict object)")
