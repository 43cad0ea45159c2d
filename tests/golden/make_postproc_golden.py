#!/usr/bin/env python3
"""Golden vectors for SURVEY row f2 made by THE REFERENCE'S OWN CODE (run in the build container only).

    python tests/golden/make_postproc_golden.py        ->  tests/golden/postproc_golden.npz

Imports /root/reference/DriveSceneGen/vectorization/utils/image_utils.py by path (it needs numpy, PIL and
matplotlib only -- all installed; matplotlib is put on the Agg backend first) and calls its
``get_gray_image(Image)`` on seeded uint8 scenes.  What is stored: the inputs and the [H, W] masks (channel 0
of the 3x stacked image the reference returns).  No reference source text is stored.

Cases (each RGB uint8):
  0-3  synthetic scene rasters (flat background + lanes + agents), as the sampler produces them
  4    a TIE in the histogram peak of channel 0 (two byte values with the same count: argmax takes the first)
  5    values at exactly +-0.1 of the peak's bin edge and one byte either side of it (the <= 0.1 decision
       in float64 on u/255 against a bin EDGE k/256 -- the comparison that a float32 or integer shortcut
       gets wrong)
  6    peak in the last bin (byte 255 falls in bin 255, whose left edge is 255/256)
  7    uniform random bytes (no dominant peak)

NOT covered by the reference here: vectorization/direct/extract_vehicles.py:136-148 (the agent threshold)
imports cv2, which this image does not have; oracle/postproc_oracle.agent_threshold stays a restatement.
"""
import importlib.util
import os
import sys

import matplotlib
matplotlib.use("Agg")
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/DriveSceneGen/vectorization/utils/image_utils.py"


def reference_module():
    spec = importlib.util.spec_from_file_location("ref_image_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def scene(seed, h=96, w=80):
    """Flat background (one byte value per channel), lanes / agents in other values, light noise."""
    rng = np.random.default_rng(seed)
    img = np.empty((h, w, 3), np.uint8)
    img[...] = rng.integers(90, 170, 3, dtype=np.uint8)
    for _ in range(6):
        r0, c0 = rng.integers(0, h - 8), rng.integers(0, w - 8)
        img[r0:r0 + rng.integers(2, 8), :, :] = rng.integers(0, 256, 3, dtype=np.uint8)
        img[:, c0:c0 + rng.integers(2, 6), :] = rng.integers(0, 256, 3, dtype=np.uint8)
    speck = rng.random((h, w)) < 0.05
    img[speck] = rng.integers(0, 256, (int(speck.sum()), 3), dtype=np.uint8)
    return img


def cases():
    out = [scene(100 + i) for i in range(4)]
    # 4: tie in channel 0's peak (values 60 and 200, the same count), channel 1 a clear peak at 128
    t = np.zeros((64, 64, 3), np.uint8)
    t[:32, :, 0], t[32:, :, 0] = 200, 60
    t[:, :, 1] = 128
    t[::7, ::5, 1] = 140
    t[:, :, 2] = 9
    out.append(t)
    # 5: every byte value once per row in channels 0 / 1 around a dominant peak at byte 128 (edge 128/256 = 0.5):
    #    u/255 within 0.1 of 0.5 <=> u in [102, 153]; 102/255 = 0.4 exactly on the boundary in float64 terms
    e = np.full((64, 256, 3), 128, np.uint8)
    e[0, :, 0] = np.arange(256)
    e[1, :, 1] = np.arange(256)
    e[2, :, 0] = np.arange(256)
    e[2, :, 1] = np.arange(256)[::-1]
    out.append(e)
    # 6: peak in the last bin
    l = np.full((48, 48, 3), 255, np.uint8)
    l[::3, ::4, 0] = 228
    l[1::5, ::3, 1] = 230
    l[::2, 1::7, :] = 229
    out.append(l)
    # 7: uniform noise
    out.append(np.random.default_rng(7).integers(0, 256, (64, 72, 3), dtype=np.uint8))
    return out


def main():
    if not os.path.exists(REF):
        sys.exit(f"{REF} is not here: this script runs in the build container only")
    ref = reference_module()
    store = {}
    for i, img in enumerate(cases()):
        gray = np.asarray(ref.get_gray_image(Image.fromarray(img)))
        assert gray.shape == img.shape and gray.dtype == np.uint8
        assert np.array_equal(gray[:, :, 0], gray[:, :, 1]) and np.array_equal(gray[:, :, 0], gray[:, :, 2])
        store[f"img{i}"] = img
        store[f"mask{i}"] = gray[:, :, 0]
    store["n"] = np.int64(len(cases()))
    path = os.path.join(HERE, "postproc_golden.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, {k: v.shape for k, v in store.items() if k.startswith("mask")})


if __name__ == "__main__":
    main()
