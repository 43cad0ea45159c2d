"""Deterministic, torch-RNG-independent synthetic weights and scene rasters (numpy only).

There is no network for checkpoints or the Waymo-derived dataset, so benchmarks and fixtures use
 - counter-based weights scaled like torch's default inits (kaiming-uniform(a=sqrt 5) for conv/linear,
   i.e. U(-1/sqrt(fan_in), 1/sqrt(fan_in)); GroupNorm affine perturbed away from (1, 0) so that the
   norm statistics matter), and
 - rasters mimicking what DriveSceneGen trains on: lane centre-lines coloured by normalised (dx, dy)
   on a grey 0.5 background in channels 0/1 and agent boxes whose value encodes speed on black in
   channel 2 (reference: DriveSceneGen/utils/datasets/rasterization.py:174-187,
   DriveSceneGen/utils/datasets/visualization.py:229,296,328), after dataset.py:21-24's [-1,1] map.
"""
from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n float64 uniforms in [0,1) from counter-based splitmix64(seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
                           ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)))
        idx = np.arange(n, dtype=np.uint64)
        z = _splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + base)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, shape, stream: int = 0) -> np.ndarray:
    """Standard normals (Box-Muller on the counter-based uniforms), float32."""
    n = int(np.prod(shape))
    m = (n + 1) // 2
    u1 = uniform01(seed, m, stream * 2 + 1)
    u2 = uniform01(seed, m, stream * 2 + 2)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
    return z.astype(np.float32).reshape(shape)


def _name_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) << 20) ^ seed


def synth_state_dict(shapes: dict, seed: int = 14555) -> dict:
    """name -> float32 ndarray for every (name, shape) of a UNet2DModel state dict."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        n = int(np.prod(shape))
        u = uniform01(_name_seed(name, seed), n)
        leaf = name.rsplit(".", 2)
        is_norm = "norm" in leaf[-2]
        if is_norm:
            v = (1.0 + 0.2 * (u - 0.5)) if name.endswith("weight") else 0.2 * (u - 0.5)
        else:
            if name.endswith("weight"):
                fan_in = int(np.prod(shape[1:]))
            else:  # bias: fan_in of the matching weight is not known here; use a small fixed bound
                fan_in = 64
            bound = 1.0 / np.sqrt(fan_in)
            v = (2.0 * u - 1.0) * bound
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def synth_scene_rasters(batch: int, channels: int, height: int, width: int, seed: int = 14555) -> np.ndarray:
    """[B,C,H,W] float32 in [-1,1]: map channels (background 0) with poly-lines, agent channels
    (background -1) with rotated boxes."""
    agent_ch = {3: (2,), 4: (2, 3), 8: (2, 7)}.get(channels, (channels - 1,))
    x = np.zeros((batch, channels, height, width), np.float32)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    for b in range(batch):
        u = uniform01(seed, 4096, stream=1000 + b)
        k = 0
        for c in agent_ch:
            x[b, c] = -1.0
        map_ch = [c for c in range(channels) if c not in agent_ch]
        for _ in range(20):  # lanes: a few straight segments chained, 2-3 px wide
            px, py = u[k] * width, u[k + 1] * height
            ang = u[k + 2] * 2 * np.pi
            k += 3
            vals = [2.0 * (u[k + i] * 0.99) - 1.0 for i in range(len(map_ch))]
            k += len(map_ch)
            for _seg in range(3):
                ln = 20 + 60 * u[k] * max(height, width) / 256.0
                ang += (u[k + 1] - 0.5) * 0.8
                k += 2
                qx, qy = px + ln * np.cos(ang), py + ln * np.sin(ang)
                dx, dy = qx - px, qy - py
                t = np.clip(((xx - px) * dx + (yy - py) * dy) / (dx * dx + dy * dy + 1e-6), 0, 1)
                d2 = (xx - (px + t * dx)) ** 2 + (yy - (py + t * dy)) ** 2
                m = d2 <= 1.5 ** 2
                for ci, c in enumerate(map_ch):
                    x[b, c][m] = vals[ci]
                px, py = qx, qy
        for _ in range(10):  # agents: rotated 10x5 px boxes, value = 2*(v/60+0.5)-1, v ~ U(0,20)
            cx, cy, ang, v = u[k] * width, u[k + 1] * height, u[k + 2] * 2 * np.pi, u[k + 3] * 20.0
            k += 4
            rx = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
            ry = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
            m = (np.abs(rx) <= 5.0) & (np.abs(ry) <= 2.5)
            for c in agent_ch:
                x[b, c][m] = 2.0 * (v / 60.0 + 0.5) - 1.0
    return x
