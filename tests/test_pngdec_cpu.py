"""The feeder's native batch PNG decoder (csrc/pngdec.hip: host code, runs without a GPU) against PIL -- the reference's
decoder (utils/datasets/dataset.py:43-45 ``ToTensor(Image.open(f))``): rows equal ``np.asarray(Image.open(f))`` bit for bit on
every 8-bit mode PIL writes, every scan-line filter, any thread count; unsupported variants and damaged files get a per-file
status and the loader reads THOSE with PIL."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest
from PIL import Image


def _decode(lib, files, h, w, c, threads):
    n = len(files)
    out = np.full((n, h, w, c), 0xAB, np.uint8)
    paths = (C.c_char_p * n)(*[f.encode() for f in files])
    status = (C.c_int32 * n)()
    rc = lib.dsg_png_decode_batch(paths, n, out.ctypes.data, h, w, c, threads, status)
    assert rc == 0, lib.dsg_last_error()
    return out, list(status)


def _png_with_filter(path, img, ftype):
    """A PNG whose every scan line uses ONE filter type (PIL's encoder picks adaptively; this forces each of the five)."""
    h, w, c = img.shape
    bpp = c
    rows = []
    prev = np.zeros(w * c, np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ftype == 0:
            f = cur
        elif ftype == 1:
            f = cur - left
        elif ftype == 2:
            f = cur - prev
        elif ftype == 3:
            f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            f = cur - pred
        rows.append(bytes([ftype]) + (f & 0xFF).astype(np.uint8).tobytes())
        prev = cur
    raw = zlib.compress(b"".join(rows), 6)

    def chunk(tag, data):
        import struct
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    import struct
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    # two IDAT chunks: the decoder must concatenate them
    half = len(raw) // 2
    blob = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"tEXt", b"k\x00v")
            + chunk(b"IDAT", raw[:half]) + chunk(b"IDAT", raw[half:]) + chunk(b"IEND", b""))
    with open(path, "wb") as f:
        f.write(blob)


@pytest.mark.parametrize("mode,c", [("L", 1), ("LA", 2), ("RGB", 3), ("RGBA", 4)])
def test_native_decoder_equals_pil_on_every_8bit_mode(lib_built, tmp_path, mode, c):
    from drivescenegen_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(c)
    h, w = 67, 45     # odd sizes: no alignment luck
    files, want = [], []
    for i in range(9):
        a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        if i % 3 == 0:   # smooth content: PIL's encoder then uses Sub / Up / Paeth lines
            a = (np.add.outer(np.arange(h), np.arange(w))[:, :, None] * (i + 1) + np.arange(c) * 40).astype(np.uint8)
        img = Image.fromarray(a[:, :, 0] if c == 1 else a, mode=mode)
        p = str(tmp_path / f"{i}.png")
        img.save(p, optimize=bool(i % 2))
        files.append(p)
        got = np.asarray(Image.open(p))
        want.append(got[:, :, None] if got.ndim == 2 else got)
    hh, ww, cc = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.dsg_png_probe(files[0].encode(), C.byref(hh), C.byref(ww), C.byref(cc)) == 0
    assert (hh.value, ww.value, cc.value) == (h, w, c)
    for threads in (1, 4, 32):
        out, status = _decode(lib, files, h, w, c, threads)
        assert status == [0] * 9
        assert np.array_equal(out, np.stack(want))


@pytest.mark.parametrize("ftype", [0, 1, 2, 3, 4])
def test_each_scanline_filter_and_split_idat(lib_built, tmp_path, ftype):
    from drivescenegen_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(10 + ftype)
    for c in (1, 3, 4):
        a = rng.integers(0, 256, (19, 23, c), dtype=np.uint8)
        p = str(tmp_path / f"f{ftype}_{c}.png")
        _png_with_filter(p, a, ftype)
        pil = np.asarray(Image.open(p))
        assert np.array_equal(pil.reshape(a.shape), a)              # (the hand-made file is a PNG PIL reads as `a`)
        out, status = _decode(lib, [p], 19, 23, c, 1)
        assert status == [0] and np.array_equal(out[0], a)


def test_unsupported_and_damaged_files_get_a_status_and_leave_the_other_rows_right(lib_built, tmp_path):
    from drivescenegen_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    good = str(tmp_path / "good.png")
    Image.fromarray(a).save(good)
    pal = str(tmp_path / "pal.png")
    Image.fromarray(a).convert("P").save(pal)
    deep = str(tmp_path / "deep.png")
    Image.fromarray(rng.integers(0, 65536, (16, 16), dtype=np.uint16)).save(deep)
    other = str(tmp_path / "other.png")
    Image.fromarray(rng.integers(0, 256, (8, 16, 3), dtype=np.uint8)).save(other)
    cut = str(tmp_path / "cut.png")
    blob = open(good, "rb").read()
    open(cut, "wb").write(blob[:len(blob) // 2])
    text = str(tmp_path / "text.png")
    open(text, "wb").write(b"not a png at all, but long enough to hold a header .........")
    missing = str(tmp_path / "missing.png")
    out, status = _decode(lib, [good, pal, deep, other, cut, text, missing, good], 16, 16, 3, 3)
    assert status == [0, 3, 3, 4, 5, 2, 1, 0]
    assert np.array_equal(out[0], a) and np.array_equal(out[7], a)
    h, w, c = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.dsg_png_probe(pal.encode(), C.byref(h), C.byref(w), C.byref(c)) == -2      # DSG_ERR_UNSUPPORTED_SHAPE
    assert lib.dsg_png_probe(missing.encode(), C.byref(h), C.byref(w), C.byref(c)) == -1
    assert lib.dsg_png_decode_batch(None, 1, None, 1, 1, 1, 1, None) == -1


def test_loader_host_side_native_pool_and_pil_give_the_same_batches(lib_built, tmp_path):
    """``GpuImageLoader._decode`` (the host half of the loader; no GPU needed): the native path, the PIL pool and a batch with a
    palette file in it (native for the rest, PIL for that row) all stage the arrays PIL gives, in the sampler's order."""
    from drivescenegen_amd import imageops
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, (10, 24, 20, 3), dtype=np.uint8)
    for i in range(10):
        # file 04 carries an ancillary tRNS chunk (still colour type 2: PIL keeps mode RGB, the decoder skips the chunk)
        Image.fromarray(imgs[i]).save(str(tmp_path / f"{i:02d}.png"), **({"transparency": (1, 2, 3)} if i == 4 else {}))
    ids = [7, 4, 0, 9, 2]
    for kw in (dict(native_png=True, workers=3), dict(native_png=False, workers=3), dict(native_png=False, workers=1)):
        ld = imageops.GpuImageLoader(str(tmp_path / "*.png"), (16, 16), batch_size=5, shuffle=False, device="cpu", **kw)
        slot, rows = ld._decode(ids)
        assert rows == 5 and np.array_equal(slot["buf"].numpy()[:5], imgs[ids]), kw
        assert ld.decode_rate(2) > 0
    # a batch holding a file only PIL reads: a ".png" that is really a BMP of the batch's shape
    bmp = str(tmp_path / "10.png")
    Image.fromarray(imgs[3]).save(bmp, format="BMP")
    ld = imageops.GpuImageLoader(sorted(str(p) for p in tmp_path.glob("*.png")), (16, 16), batch_size=3, shuffle=False,
                                 device="cpu", workers=2)
    slot, rows = ld._decode([1, 10, 5])       # row 1 of the batch: status 2 (not a PNG) -> PIL
    assert np.array_equal(slot["buf"].numpy()[:3], imgs[[1, 3, 5]])
    # staging slots are reused, not re-allocated: the ring has prefetch + 2 of them
    ptrs = {ld._decode([0, 1, 2])[0]["buf"].data_ptr() for _ in range(12)}
    assert len(ptrs) == ld.prefetch + 2


def test_corrupt_pickle_raises_and_refused_pickle_warns(tmp_path):
    """only the restricted unpickler's refusal makes a .pkl skippable (with a warning that names the file); a
    truncated file raises instead of being silently replaced by its neighbour."""
    import pickle
    import torch
    from drivescenegen_amd.imageops import load_sample_pickle
    good = tmp_path / "a.pkl"
    torch.save({"fig_tensor": torch.zeros(4, 4, 3)}, good)
    assert isinstance(load_sample_pickle(str(good)), dict)
    cut = tmp_path / "b.pkl"
    cut.write_bytes(good.read_bytes()[:40])
    with pytest.raises(Exception) as ei:
        load_sample_pickle(str(cut))
    assert ei.value is not None     # (whatever torch.load raises for a truncated zip: it propagates, the neighbour is NOT used)

    class Evil:
        def __reduce__(self):
            return (os.getcwd, ())
    bad = tmp_path / "c.pkl"
    torch.save({"fig_tensor": Evil()}, bad)
    with pytest.warns(RuntimeWarning, match="trust_pickles"):
        assert load_sample_pickle(str(bad)) is None
