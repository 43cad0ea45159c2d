"""Numpy restatement of the engine's counter-based device noise -- TEST INFRASTRUCTURE (oracle).

What it checks: ``dsg_philox_u32`` / ``dsg_philox_normal`` / ``dsg_add_noise_philox`` (include/dsg.h), the opt-in
replacement of the reference's host draw at /root/reference/DriveSceneGen/pipeline/training_pipeline.py:72
(``noise = torch.randn(batch.shape).to(device)``) fused with ``add_noise`` (:80).  The reference has no device generator
of its own; the algorithm is a PUBLISHED one, restated here from the paper:

    J. K. Salmon, M. A. Moraes, R. O. Dror, D. E. Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11 --
    Philox4x32-10: 10 rounds of   (c0, c1, c2, c3) <- (mulhi(M1, c2) ^ c1 ^ k0,  mullo(M1, c2),
                                                      mulhi(M0, c0) ^ c3 ^ k1,  mullo(M0, c0)),
    M0 = 0xD2511F53, M1 = 0xCD9E8D57, the key bumped by (0x9E3779B9, 0xBB67AE85) between rounds.

Pinned by the Random123 distribution's known-answer vectors (``kat_vectors``: philox4x32 10), see
tests/test_oracle_kat.py::test_philox_oracle_known_answers.  The stream layout and the Box-Muller step are this
project's own definition (drivescenegen_amd/csrc/scheduler.hip, the comment above ``philox4x32_10``):

    element e of the flat tensor = lane e % 4 of block  philox(counter = (c lo, c hi, offset lo, offset hi), key = (seed lo,
    seed hi)),  c = e // 4;   lanes (0, 1) and (2, 3) are Box-Muller pairs
        u1 = (fp32(r_even) + 0.5) * 2^-32     (each operation rounded to fp32)
        u2 =  fp32(r_odd) * 2^-32
        z_even = sqrt(-2 ln u1) cos(2 pi u2),   z_odd = sqrt(-2 ln u1) sin(2 pi u2)

``normals`` evaluates log / sqrt / cos / sin in float64 on the fp32 u1, u2: the device's fp32 libm differs from it by
rounding only (the GPU test's tolerance says how much).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: uint32 [..., 4], key: uint32 [..., 2] (broadcastable) -> uint32 [..., 4]."""
    c = [np.asarray(counter)[..., i].astype(np.uint64) for i in range(4)]
    k = [np.asarray(key)[..., i].astype(np.uint64) for i in range(2)]
    for rnd in range(10):
        p0 = M0 * c[0]          # 32 x 32 -> 64 bits, exact in uint64
        p1 = M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c = [hi1 ^ c[1] ^ k[0], lo1, hi0 ^ c[3] ^ k[1], lo0]
        if rnd < 9:
            k = [(k[0] + np.uint64(W0)) & _MASK, (k[1] + np.uint64(W1)) & _MASK]
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def stream_u32(numel: int, seed: int, offset: int) -> np.ndarray:
    """The first `numel` uint32 of the tensor named (seed, offset)."""
    blocks = (numel + 3) // 4
    c = np.arange(blocks, dtype=np.uint64)
    ctr = np.stack([(c & _MASK), (c >> np.uint64(32)), np.full(blocks, offset & 0xFFFFFFFF, np.uint64),
                    np.full(blocks, (offset >> 32) & 0xFFFFFFFF, np.uint64)], axis=-1).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return philox4x32_10(ctr, key).reshape(-1)[:numel]


def uniforms(r_even: np.ndarray, r_odd: np.ndarray):
    """The two fp32 uniforms of a Box-Muller pair, each operation rounded to fp32 as on the device."""
    u1 = (r_even.astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -32)
    u2 = r_odd.astype(np.float32) * np.float32(2.0 ** -32)
    return u1, u2


def normals(numel: int, seed: int, offset: int) -> np.ndarray:
    """float64 [numel]: Box-Muller of the stream, transcendental functions in float64."""
    r = stream_u32((numel + 3) // 4 * 4, seed, offset).reshape(-1, 2)
    u1, u2 = uniforms(r[:, 0], r[:, 1])
    rad = np.sqrt(-2.0 * np.log(u1.astype(np.float64)))
    th = 2.0 * np.pi * u2.astype(np.float64)
    return np.stack([rad * np.cos(th), rad * np.sin(th)], axis=-1).reshape(-1)[:numel]


def add_noise(x0: np.ndarray, sqrt_a: np.ndarray, sqrt_1ma: np.ndarray, noise: np.ndarray) -> np.ndarray:
    """x_t = sqrt_a[n] * x0 + sqrt_1ma[n] * noise with two multiplies and one add, each rounded to fp32."""
    shape = (-1,) + (1,) * (x0.ndim - 1)
    a, b = sqrt_a.astype(np.float32).reshape(shape), sqrt_1ma.astype(np.float32).reshape(shape)
    return (a * x0.astype(np.float32)).astype(np.float32) + (b * noise.astype(np.float32)).astype(np.float32)
