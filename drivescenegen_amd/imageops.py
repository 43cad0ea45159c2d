"""GPU input pipeline and output post-processing around the denoising path.

f1  ``GpuImageLoader``: the reference feeds training from a single-threaded PIL/torchvision dataset
    (utils/datasets/dataset.py:15-50 with DataLoader(num_workers=0), scripts/train.py:35).  Here the host only
    decodes PNGs into pinned uint8 staging buffers (a prefetch thread), copies them on a side stream, and ONE
    kernel does ToTensor + bilinear Resize(antialias=False) + Normalize for the whole batch.
f2  ``gray_mask_batch`` / ``agent_mask_batch``: get_gray_image's histogram-peak background detection and +-0.1
    mask (vectorization/utils/image_utils.py:13-43, its per-pixel Python loop at :40 is the vectoriser's first
    bottleneck) and extract_agents' threshold (vectorization/direct/extract_vehicles.py:136-148) for a batch of
    generated images that is still on the GPU.
"""
from __future__ import annotations

import glob
import queue
import threading

import numpy as np
import torch

from . import _lib


def resize_normalize(images_u8: torch.Tensor, size, mean=0.5, std=0.5) -> torch.Tensor:
    """uint8 [N, Hs, Ws, C] (GPU) -> fp32 [N, C, H, W] = Normalize(Resize(ToTensor(img)))."""
    if not images_u8.is_cuda or images_u8.dtype != torch.uint8:
        raise RuntimeError("resize_normalize: expects a uint8 GPU tensor [N, H, W, C] (no CPU fallback)")
    n, hs, ws, c = images_u8.shape
    out = torch.empty((n, c, size[0], size[1]), dtype=torch.float32, device=images_u8.device)
    x = images_u8.contiguous()
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_resize_normalize_u8(x.data_ptr(), n, hs, ws, c, _lib.ptr(out), size[0], size[1],
                                                      float(mean), float(std), _lib.stream_ptr(x.device)))
    return out


class GpuImageLoader:
    """Iterates batches of normalised fp32 [B, C, H, W] GPU tensors from image files.  Host threads decode and
    pin; the H2D copy runs on a side stream two batches ahead; resize + normalise is one HIP kernel."""

    def __init__(self, pattern_or_files, size, batch_size, shuffle=True, seed=0, device="cuda", prefetch=2,
                 rank=0, world=1, drop_last=False):
        self.files = sorted(glob.glob(pattern_or_files)) if isinstance(pattern_or_files, str) else list(
            pattern_or_files)
        self.size, self.bs, self.shuffle, self.seed = tuple(size), batch_size, shuffle, seed
        self.device = torch.device(device)
        self.prefetch, self.rank, self.world, self.drop_last = prefetch, rank, world, drop_last
        self.epoch = 0
        self._copy_stream = None

    def __len__(self):
        nb = len(self.files) // self.bs if self.drop_last else -(-len(self.files) // self.bs)
        return -(-nb // self.world)

    def _batches(self):
        idx = np.arange(len(self.files))
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(idx)
        bl = [idx[i:i + self.bs] for i in range(0, len(idx), self.bs)]
        if self.drop_last and bl and len(bl[-1]) < self.bs:
            bl.pop()
        return [b for i, b in enumerate(bl) if i % self.world == self.rank]

    def _decode(self, ids):
        from PIL import Image
        arrs = []
        for i in ids:
            a = np.asarray(Image.open(self.files[i]))
            arrs.append(a[:, :, None] if a.ndim == 2 else a)
        if any(a.shape != arrs[0].shape for a in arrs):
            raise ValueError("GpuImageLoader: images of one batch must share a shape")
        host = torch.from_numpy(np.stack(arrs)).pin_memory()
        return host

    def __iter__(self):
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        batches = self._batches()
        self.epoch += 1

        def producer():
            try:
                for ids in batches:
                    q.put(self._decode(ids))
            except Exception as e:  # surfaced in the consumer
                q.put(e)
            q.put(None)

        threading.Thread(target=producer, daemon=True).start()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(self.device)
        while True:
            host = q.get()
            if host is None:
                return
            if isinstance(host, Exception):
                raise host
            with torch.cuda.stream(self._copy_stream):
                dev = host.to(self.device, non_blocking=True)
            torch.cuda.current_stream(self.device).wait_stream(self._copy_stream)
            dev.record_stream(torch.cuda.current_stream(self.device))
            yield resize_normalize(dev, self.size)


# --------------------------------------------------------------------------------------------------
# f2: output post-processing
# --------------------------------------------------------------------------------------------------
_U = np.arange(256)
# np.histogram(bins=256, range=(0,1)) bin of the value u/255 (float64), for every byte value u
_BIN_OF_U = np.array([int(np.argmax(np.histogram(np.array([u / 255.0]), bins=256, range=(0, 1))[0])) for u in _U])
_EDGES = np.histogram(np.array([0.0]), bins=256, range=(0, 1))[1]


def histograms_u8(images_u8: torch.Tensor) -> torch.Tensor:
    """uint8 [N, H, W, C] (GPU) -> int64 [N, C, 256] byte histograms."""
    n, h, w, c = images_u8.shape
    hist = torch.empty((n, c, 256), dtype=torch.int32, device=images_u8.device)
    x = images_u8.contiguous()
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_hist_u8(x.data_ptr(), n, h * w, c, hist.data_ptr(), _lib.stream_ptr(x.device)))
    return hist


def _mask_lut(images_u8, luts, ch0, ch1, on, off):
    n, h, w, c = images_u8.shape
    lut = torch.from_numpy(np.ascontiguousarray(luts, dtype=np.uint8)).to(images_u8.device)
    out = torch.empty((n, h, w), dtype=torch.uint8, device=images_u8.device)
    x = images_u8.contiguous()
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dsg_mask_lut_u8(x.data_ptr(), n, h * w, c, ch0, ch1, lut.data_ptr(), on, off,
                                              out.data_ptr(), _lib.stream_ptr(x.device)))
    return out


def gray_mask_batch(images_u8: torch.Tensor) -> torch.Tensor:
    """get_gray_image for a batch: uint8 [N, H, W, >=2] (GPU) -> uint8 [N, H, W] mask (0 = background, 255 = lane).
    The histogram runs on the GPU; the 256-entry decision tables are built on the host with the reference's float64
    arithmetic (peak bin edge mx, |u/255 - mx| <= 0.1), so the mask is bit-identical to the reference loop."""
    if not images_u8.is_cuda or images_u8.dtype != torch.uint8:
        raise RuntimeError("gray_mask_batch: expects a uint8 GPU tensor [N, H, W, C] (no CPU fallback)")
    hist_u = histograms_u8(images_u8)[:, :2].cpu().numpy().astype(np.int64)  # [N, 2, 256]
    n = hist_u.shape[0]
    luts = np.zeros((n, 2, 256), np.uint8)
    vals = _U / 255.0
    for i in range(n):
        for ch in range(2):
            hb = np.bincount(_BIN_OF_U, weights=hist_u[i, ch], minlength=256)
            m = _EDGES[int(np.argmax(hb))]
            luts[i, ch] = np.fabs(vals - m) <= 0.1
    return _mask_lut(images_u8, luts, 0, 1, 0, 255)  # both channels near the peak -> background (0)


def agent_mask_batch(images_u8: torch.Tensor, channel: int = 2, thresh: int = 100) -> torch.Tensor:
    """extract_agents' binarisation for a batch of generated uint8 images [N, H, W, C] (GPU): the reference
    re-reads the PNG as float32 /255, multiplies by 255, truncates to uint8 and thresholds at 100."""
    if not images_u8.is_cuda or images_u8.dtype != torch.uint8:
        raise RuntimeError("agent_mask_batch: expects a uint8 GPU tensor [N, H, W, C] (no CPU fallback)")
    f = (_U.astype(np.float32) / np.float32(255)).astype(np.float32)
    lut1 = ((f * 255).astype(np.uint8) > thresh).astype(np.uint8)
    luts = np.broadcast_to(np.stack([lut1, lut1])[None], (images_u8.shape[0], 2, 256))
    return _mask_lut(images_u8, luts, channel, -1, 255, 0)
