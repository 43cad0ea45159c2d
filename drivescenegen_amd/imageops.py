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


_warned_pickles = set()


def load_sample_pickle(path, trust_pickles=False):
    """A ``.pkl`` sample of the reference's dataset (utils/datasets/dataset.py:31-41): ``{"fig_tensor": tensor}``.  Read with
    ``torch.load(weights_only=True)`` -- such a dict needs nothing more, and a data directory is not a place to run code from.
    ONLY the restricted unpickler's refusal of a global it does not allow (``pickle.UnpicklingError``) makes a file "not a
    usable dict" (returns None, with one warning per file that names it and ``trust_pickles=True``): the reference skips a
    non-dict pickle for the next file (dataset.py:37-39) and so do the callers.  A truncated or corrupt file, an I/O error
    -- anything else -- propagates: the reference's ``torch.load`` would raise there too, and silently training on a
    neighbour instead would duplicate samples.  ``trust_pickles=True`` falls back to full unpickling -- which executes code
    from the file -- for pickles that need it."""
    import pickle
    with open(path, "rb") as fh:
        try:
            return torch.load(fh, weights_only=True)
        except pickle.UnpicklingError as e:
            if not trust_pickles:
                if path not in _warned_pickles:
                    _warned_pickles.add(path)
                    import warnings
                    warnings.warn(f"{path}: skipped -- the restricted unpickler refused it ({str(e).splitlines()[0][:120]}); "
                                  "pass trust_pickles=True to unpickle it fully (that executes code from the file)",
                                  RuntimeWarning, stacklevel=2)
                return None
            fh.seek(0)
            return torch.load(fh, weights_only=False)


class GpuImageLoader:
    """Iterates batches of normalised fp32 [B, C, H, W] GPU tensors from the files ``Image_Dataset`` reads
    (utils/datasets/dataset.py:15-50): PNG / any PIL image, or ``.pkl`` with a ``fig_tensor`` [H, W, C] float entry
    (a pickle that is not a dict falls through to the next file, dataset.py:39-40).

    Host side: a batch's files are decoded in parallel, each image straight into its row of a pinned staging buffer from a
    small ring that is allocated once and reused (a slot is rewritten only after the H2D copy that read it has completed:
    an event per slot); batches come out in the sampler's order whatever order the decodes finish in.  The H2D copy and
    the one resize + normalise kernel of a batch are issued by the producer thread on a side stream as soon as the batch is
    decoded, `prefetch` batches ahead of the consumer.  PNG batches go through the library's own decoder
    (``dsg_png_decode_batch``, csrc/pngdec.hip: `workers` native threads, no GIL); everything else -- other formats, .pkl
    samples, PNG variants the decoder leaves alone -- through a pool of `workers` Python threads over PIL (which scales
    poorly: its chunk loop holds the GIL).  One PIL thread gives 145-326 images/s on 512x512 RGB PNGs -- below what a
    training step consumes (249 images/s fp32, 699 bf16); ``decode_rate()`` measures the pool (5800-6200 images/s with 16
    workers on the GPU box).

    Under data parallelism (`rank`, `world`) it follows accelerate's prepared loader exactly like
    ``training._ShardedLoader`` (sharding.py): ONE shuffled order per epoch on every rank -- rank 0 draws the epoch's
    seed, broadcasts it when a process group exists (`seed=None`), or all ranks derive it from `seed + epoch` -- rank r
    takes batches r, r + W, ..., and a ragged tail is completed from the start of the order so that every rank runs
    ``len(loader)`` full-size steps (a rank short of one batch would never join the last gradient all-reduce)."""

    def __init__(self, pattern_or_files, size, batch_size, shuffle=True, seed=0, device="cuda", prefetch=2,
                 rank=0, world=1, drop_last=False, trust_pickles=False, workers=None, native_png=True):
        self.files = sorted(glob.glob(pattern_or_files)) if isinstance(pattern_or_files, str) else list(
            pattern_or_files)
        self.size, self.bs, self.shuffle, self.seed = tuple(size), batch_size, shuffle, seed
        self.device = torch.device(device)
        self.prefetch, self.rank, self.world, self.drop_last = prefetch, rank, world, drop_last
        # .pkl samples are read with torch.load(weights_only=True) (a {"fig_tensor": tensor} dict needs nothing more);
        # trust_pickles=True falls back to full unpickling -- which runs code from the data directory -- for other pickles
        self.trust_pickles = bool(trust_pickles)
        import os
        self.workers = max(1, int(workers)) if workers is not None else max(1, min(16, (os.cpu_count() or 2) - 1))
        self.native_png = bool(native_png)   # False: every file through PIL (the A/B switch of decode_rate)
        self.epoch = 0
        self.epoch_seed = None
        self._copy_stream = None
        self._pool = None
        self._ring = None        # [{"buf": pinned tensor [B, H, W, C], "event": copy-finished event or None}] * (prefetch + 2)
        self._ring_key = None
        self._slot = 0

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
                with Image.open(f) as im:
                    a = np.asarray(im)
                return a[:, :, None] if a.ndim == 2 else a
            dd = load_sample_pickle(f, self.trust_pickles)
            if isinstance(dd, dict):
                return np.ascontiguousarray(dd["fig_tensor"][:, :, :].float().numpy())
        raise IndexError(f"GpuImageLoader: no usable sample among {len(self.files)} files (every .pkl was refused by the "
                         "restricted unpickler or holds a non-dict object; see trust_pickles)")

    # ---- decode pool + pinned staging ring -----------------------------------------------------------------------
    def _executor(self):
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="dsg-decode")
        return self._pool

    def _staging(self, shape, dtype):
        """The next slot of the ring for batches of `shape` / `dtype` (re-made when a batch of another shape arrives);
        waits for the H2D copy that last read the slot."""
        key = (tuple(shape[1:]), dtype)
        if self._ring is None or self._ring_key != key:
            pin = torch.cuda.is_available()
            tdt = torch.uint8 if dtype == np.uint8 else torch.float32
            self._ring = [{"buf": torch.empty((self.bs,) + key[0], dtype=tdt, pin_memory=pin), "event": None}
                          for _ in range(self.prefetch + 2)]
            self._ring_key, self._slot = key, 0
        slot = self._ring[self._slot]
        self._slot = (self._slot + 1) % len(self._ring)
        if slot["event"] is not None:
            slot["event"].synchronize()
            slot["event"] = None
        return slot

    def _decode_native(self, ids):
        """A batch of .png files through the library's native decoder (``dsg_png_decode_batch``: `workers` C++ threads, the
        whole call outside the GIL, rows written straight into the pinned slot).  Files it does not take (16-bit, palette,
        interlaced, damaged: a per-file status) are read with PIL into the same rows.  None: not a pure-PNG batch, or the
        first file is not one the decoder handles -- the generic pool path decodes the batch."""
        import ctypes as C
        files = [self.files[i] for i in ids]
        if not self.native_png or not all(f.lower().endswith(".png") for f in files):
            return None
        lib = _lib.load()
        h, w, c = C.c_int32(), C.c_int32(), C.c_int32()
        if lib.dsg_png_probe(files[0].encode(), C.byref(h), C.byref(w), C.byref(c)) != 0:
            return None
        n = len(files)
        slot = self._staging((n, h.value, w.value, c.value), np.uint8)
        paths = (C.c_char_p * n)(*[f.encode() for f in files])
        status = (C.c_int32 * n)()
        _lib.check(lib.dsg_png_decode_batch(paths, n, slot["buf"].data_ptr(), h.value, w.value, c.value, self.workers, status))
        view = None
        for j, st in enumerate(status):
            if st != 0:     # this file goes through PIL (which also raises the proper error for a damaged file)
                a = self._load_one(ids[j])
                view = slot["buf"].numpy() if view is None else view
                if a.shape != view.shape[1:] or a.dtype != np.uint8:
                    raise ValueError("GpuImageLoader: images of one batch must share a shape")
                np.copyto(view[j], a)
        return slot, n

    def _decode(self, ids):
        """Decode one batch on the pool: returns (slot, rows) -- the pinned [rows, H, W, C] staging view is slot["buf"][:rows]."""
        ids = list(ids)
        native = self._decode_native(ids)
        if native is not None:
            return native
        first = self._load_one(ids[0])    # (names the batch's shape / dtype; the other members are decoded by the pool)
        mixed = {}

        def work(j, i, slot_np):
            a = first if j == 0 else self._load_one(i)
            if a.shape != slot_np.shape[1:]:
                raise ValueError("GpuImageLoader: images of one batch must share a shape")
            if a.dtype != slot_np.dtype:
                if slot_np.dtype == np.uint8:     # a .pkl member in a PNG batch: the whole batch must travel as float32
                    mixed[j] = a
                    return
                a = a.astype(np.float32) / np.float32(255) if a.dtype == np.uint8 else a.astype(np.float32, copy=False)
            np.copyto(slot_np[j], a)

        dtype = np.uint8 if first.dtype == np.uint8 else np.float32
        slot = self._staging((len(ids),) + first.shape, dtype)
        view = slot["buf"].numpy()
        list(self._executor().map(lambda ji: work(ji[0], ji[1], view), enumerate(ids)))
        if mixed:   # (rare: ToTensor's / 255 on the host for the uint8 members, as a batch of float32)
            slot = self._staging((len(ids),) + first.shape, np.float32)
            fview = slot["buf"].numpy()
            for j in range(len(ids)):
                fview[j] = mixed[j].astype(np.float32, copy=False) if j in mixed else view[j].astype(np.float32) / np.float32(255)
        return slot, len(ids)

    def decode_rate(self, batches: int = 4):
        """Images / s of the host side alone (decode pool into pinned staging, no GPU work): what the loader can feed."""
        import time
        todo = self._batches()[:batches]
        self._decode(todo[0])          # (pool start-up, ring allocation)
        t0 = time.perf_counter()
        n = sum(self._decode(ids)[1] for ids in todo)
        return n / (time.perf_counter() - t0)

    def __iter__(self):
        """The producer thread does everything that can run ahead of the consumer: decode into a pinned slot, H2D copy and
        the resize + normalise kernel on the loader's side stream, an event behind them -- `prefetch` finished batches wait
        in the queue.  The consumer only makes its stream wait for the batch's event (no copy is ever issued from the
        training thread, none waits for the training stream)."""
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        batches = self._batches()
        self.epoch += 1
        stop = threading.Event()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(self.device)
        side = self._copy_stream

        def producer():
            try:
                for ids in batches:
                    if stop.is_set():
                        return
                    slot, rows = self._decode(ids)
                    with torch.cuda.stream(side):
                        dev = slot["buf"][:rows].to(self.device, non_blocking=True)
                        copied = torch.cuda.Event()
                        copied.record(side)
                        slot["event"] = copied   # the slot is rewritten only after this copy has read it (ring of prefetch + 2)
                        out = resize_normalize(dev, self.size)
                        ready = torch.cuda.Event()
                        ready.record(side)
                    q.put((out, ready))
            except Exception as e:  # surfaced in the consumer
                q.put(e)
            q.put(None)

        worker = threading.Thread(target=producer, daemon=True)
        worker.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, Exception):
                    raise item
                out, ready = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ready)
                out.record_stream(cur)      # (allocated on the side stream, used on the consumer's)
                yield out
        finally:
            # a consumer that stops early: tell the producer, keep the queue drained so that it cannot block on a full queue, and
            # JOIN it -- when the iterator is gone no thread of this loader is still decoding, copying or launching kernels
            stop.set()
            while worker.is_alive():
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            worker.join()


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
