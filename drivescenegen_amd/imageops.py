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

from . import _lib, sharding


def resize_normalize(images: torch.Tensor, size, mean=0.5, std=0.5) -> torch.Tensor:
    """[N, Hs, Ws, C] (GPU) -> fp32 [N, C, H, W] = Normalize(Resize(.)) of
    uint8: a decoded image through ToTensor (/ 255)                         (dataset.py:43-45)
    float32: the .pkl branch's `fig_tensor`, permuted HWC -> CHW and used as is  (dataset.py:37-41)"""
    if not images.is_cuda or images.dtype not in (torch.uint8, torch.float32):
        raise RuntimeError("resize_normalize: expects a uint8 or float32 GPU tensor [N, H, W, C] (no CPU fallback)")
    n, hs, ws, c = images.shape
    out = torch.empty((n, c, size[0], size[1]), dtype=torch.float32, device=images.device)
    x = images.contiguous()
    lib = _lib.load()
    fn = lib.dsg_resize_normalize_u8 if x.dtype == torch.uint8 else lib.dsg_resize_normalize_f32
    with torch.cuda.device(x.device):
        _lib.check(fn(x.data_ptr(), n, hs, ws, c, _lib.ptr(out), size[0], size[1], float(mean), float(std),
                      _lib.stream_ptr(x.device)))
    return out


def load_sample_pickle(path, trust_pickles=False):
    """A ``.pkl`` sample of the reference's dataset (utils/datasets/dataset.py:31-41): ``{"fig_tensor": tensor}``.  Read with
    ``torch.load(weights_only=True)`` -- such a dict needs nothing more, and a data directory is not a place to run code from.
    A file that holds anything the restricted unpickler refuses is "not a usable dict" (returns None): the reference skips a
    non-dict pickle for the next file (dataset.py:37-39) and so do the callers.  ``trust_pickles=True`` falls back to full
    unpickling -- which executes code from the file -- for pickles that need it."""
    import pickle
    with open(path, "rb") as fh:
        try:
            return torch.load(fh, weights_only=True)
        except (pickle.UnpicklingError, RuntimeError, AttributeError, ImportError, EOFError, ValueError):
            if not trust_pickles:
                return None
            fh.seek(0)
            return torch.load(fh, weights_only=False)


class GpuImageLoader:
    """Iterates batches of normalised fp32 [B, C, H, W] GPU tensors from the files ``Image_Dataset`` reads
    (utils/datasets/dataset.py:15-50): PNG / any PIL image, or ``.pkl`` with a ``fig_tensor`` [H, W, C] float entry
    (a pickle that is not a dict falls through to the next file, dataset.py:39-40).  Host threads decode into pinned
    buffers; the H2D copy runs on a side stream `prefetch` batches ahead; permute + resize + normalise is one HIP kernel
    (uint8 or float source).

    Under data parallelism (`rank`, `world`) it follows accelerate's prepared loader exactly like
    ``training._ShardedLoader`` (sharding.py): ONE shuffled order per epoch on every rank -- rank 0 draws the epoch's
    seed, broadcasts it when a process group exists (`seed=None`), or all ranks derive it from `seed + epoch` -- rank r
    takes batches r, r + W, ..., and a ragged tail is completed from the start of the order so that every rank runs
    ``len(loader)`` full-size steps (a rank short of one batch would never join the last gradient all-reduce)."""

    def __init__(self, pattern_or_files, size, batch_size, shuffle=True, seed=0, device="cuda", prefetch=2,
                 rank=0, world=1, drop_last=False, trust_pickles=False):
        self.files = sorted(glob.glob(pattern_or_files)) if isinstance(pattern_or_files, str) else list(
            pattern_or_files)
        self.size, self.bs, self.shuffle, self.seed = tuple(size), batch_size, shuffle, seed
        self.device = torch.device(device)
        self.prefetch, self.rank, self.world, self.drop_last = prefetch, rank, world, drop_last
        # .pkl samples are read with torch.load(weights_only=True) (a {"fig_tensor": tensor} dict needs nothing more);
        # trust_pickles=True falls back to full unpickling -- which runs code from the data directory -- for other pickles
        self.trust_pickles = bool(trust_pickles)
        self.epoch = 0
        self.epoch_seed = None
        self._copy_stream = None

    def __len__(self):
        return sharding.steps_per_epoch(len(self.files), self.bs, self.world, self.drop_last)

    def _batches(self):
        idx = np.arange(len(self.files))
        if self.shuffle:
            self.epoch_seed = (sharding.broadcast_epoch_seed(self.rank, self.world, self.device) if self.seed is None
                               else self.seed + self.epoch)
            np.random.default_rng(self.epoch_seed).shuffle(idx)
        return sharding.shard_batches(idx.tolist(), self.bs, self.rank, self.world, self.drop_last)

    def _load_one(self, i):
        """One file as an HWC array: uint8 for images, float32 for the .pkl branch.  A .pkl that does not hold a dict is
        skipped for the next file (the reference's dataset does the same, utils/datasets/dataset.py:31-36); after one full
        round of the file list without a usable file this raises instead of recursing for ever."""
        for k in range(len(self.files)):
            f = self.files[(i + k) % len(self.files)]
            if not f.lower().endswith(".pkl"):
                from PIL import Image
                a = np.asarray(Image.open(f))
                return a[:, :, None] if a.ndim == 2 else a
            dd = load_sample_pickle(f, self.trust_pickles)
            if isinstance(dd, dict):
                return np.ascontiguousarray(dd["fig_tensor"][:, :, :].float().numpy())
        raise IndexError(f"GpuImageLoader: no usable sample among {len(self.files)} files (every .pkl holds a non-dict object)")

    def _decode(self, ids):
        arrs = [self._load_one(i) for i in ids]
        if any(a.shape != arrs[0].shape for a in arrs):
            raise ValueError("GpuImageLoader: images of one batch must share a shape")
        if any(a.dtype != np.uint8 for a in arrs):  # a batch with .pkl members travels as float32 (ToTensor's / 255 on the host)
            arrs = [a.astype(np.float32) / np.float32(255) if a.dtype == np.uint8 else a.astype(np.float32, copy=False)
                    for a in arrs]
        return torch.from_numpy(np.stack(arrs)).pin_memory()

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
