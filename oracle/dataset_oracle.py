"""Numpy restatement of the reference's dataset transform -- TEST INFRASTRUCTURE (oracle).

Follows /root/reference/DriveSceneGen/utils/datasets/dataset.py:21-24,43-45: ``ToTensor`` (uint8 HWC -> float CHW / 255),
``Resize((H, W), antialias=False)`` (torchvision on a tensor = ``F.interpolate(mode="bilinear", align_corners=False)``) and
``Normalize([0.5], [0.5])``.  torchvision is not installed here; the bilinear rule is restated from torch's definition
(``area_pixel_compute_source_index``: src = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out, neighbours i0 = floor(src),
i1 = min(i0 + 1, in - 1), weight of i1 = src - i0; fp32, the source index with ONE rounding = the fma torch's build emits) and cross-checked against ``torch.nn.functional.interpolate`` on
the CPU by tests/test_oracle_kat.py.  Parity unpinned by the reference (no tests there)."""
import numpy as np


def _axis(n_in: int, n_out: int):
    scale = np.float32(n_in) / np.float32(n_out)
    # one rounding: torch's compiled kernel contracts scale * (dst + 0.5) - 0.5 into an fma (the product is exact in float64)
    src = (scale.astype(np.float64) * (np.arange(n_out, dtype=np.float64) + 0.5) - 0.5).astype(np.float32)
    src = np.maximum(src, np.float32(0))
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, (np.float32(1) - l1).astype(np.float32), l1


def resize_bilinear(x_chw: np.ndarray, size) -> np.ndarray:
    """fp32 [C, H, W] -> [C, size[0], size[1]]."""
    x = np.asarray(x_chw, dtype=np.float32)
    y0, y1, hy0, hy1 = _axis(x.shape[1], size[0])
    x0, x1, wx0, wx1 = _axis(x.shape[2], size[1])
    top = wx0 * x[:, y0][:, :, x0] + wx1 * x[:, y0][:, :, x1]
    bot = wx0 * x[:, y1][:, :, x0] + wx1 * x[:, y1][:, :, x1]
    return (hy0[None, :, None] * top + hy1[None, :, None] * bot).astype(np.float32)


def dataset_item(img_hwc: np.ndarray, size) -> np.ndarray:
    """One sample as ``Image_Dataset.__getitem__`` returns it: uint8 HWC (a decoded image) through ToTensor, or float HWC
    (the .pkl branch's ``fig_tensor``) as it is; then Resize and Normalize(0.5, 0.5)."""
    a = np.asarray(img_hwc)
    if a.ndim == 2:
        a = a[:, :, None]
    x = a.transpose(2, 0, 1)
    x = x.astype(np.float32) / np.float32(255) if a.dtype == np.uint8 else x.astype(np.float32)
    return ((resize_bilinear(x, size) - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)
