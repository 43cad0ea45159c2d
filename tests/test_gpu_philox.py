"""The library's counter-based device noise (dsg_philox_u32 / dsg_philox_normal / dsg_add_noise_philox, csrc/scheduler.hip):
the opt-in replacement of the training loop's host draw (training_pipeline.py:72 `torch.randn(batch.shape).to(device)`) fused
with add_noise (:80).  The uint32 stream is BIT-EXACT against oracle/philox_oracle.py (pinned by the Random123 known-answer
vectors in tests/test_oracle_kat.py); the normals equal the fp64 evaluation of the same formulas to fp32 rounding (tolerance
below) and pass moment / Kolmogorov-Smirnov checks; `noisy` is bitwise add_noise(x0, noise, t) on the noise the call wrote."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import ops, synth  # noqa: E402
from oracle import philox_oracle as po  # noqa: E402

DEV = "cuda"
CASES = [(0, 0), (14555, 3), (2 ** 63 + 12345, (5 << 40) | 77), (2 ** 64 - 1, 2 ** 64 - 1)]


@pytest.mark.parametrize("seed,offset", CASES)
def test_uint32_stream_is_bit_exact(seed, offset):
    for numel in (1, 3, 4, 1021, 1 << 18):
        got = ops.philox_u32(numel, seed, offset).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, po.stream_u32(numel, seed, offset)), numel


def test_counter_carries_past_32_bits():
    """Blocks beyond 2^32 elements use the counter's second word: checked through a kernel call far into the stream is not
    possible without 16 GiB; the oracle's layout (c lo, c hi) is exercised on the device by the offset words instead -- the
    same two-word split -- and element index arithmetic is 64-bit (a [2^16 + 3] tail after a large multiple of 4)."""
    n = (1 << 22) + 3
    got = ops.philox_u32(n, 9, 1 << 33).cpu().numpy().view(np.uint32)
    want = po.stream_u32(n, 9, 1 << 33)
    assert np.array_equal(got[-4099:], want[-4099:]) and np.array_equal(got[:8], want[:8])
    assert not np.array_equal(got[:8], po.stream_u32(8, 9, 0))       # the offset's HIGH word reaches the counter


@pytest.mark.parametrize("seed,offset", CASES[:3])
def test_normals_equal_the_fp64_box_muller_to_fp32_rounding(seed, offset):
    n = (1 << 20) + 2
    got = ops.philox_normal((n,), seed, offset).cpu().numpy().astype(np.float64)
    want = po.normals(n, seed, offset)
    # fp32 logf / sqrtf / cospif / sinpif against float64: a few ulp of a value below 6.8
    assert float(np.abs(got - want).max()) <= 4e-6
    assert np.isfinite(got).all() and float(np.abs(got).max()) < 6.8
    from scipy import stats
    assert abs(got.mean()) < 5e-3 and abs(got.std() - 1) < 5e-3
    assert stats.kstest(got, "norm").pvalue > 1e-4
    assert abs(stats.skew(got)) < 1e-2 and abs(stats.kurtosis(got)) < 2e-2
    assert abs(np.corrcoef(got[:-1], got[1:])[0, 1]) < 5e-3          # Box-Muller partners are uncorrelated


@pytest.mark.parametrize("shape", [(4, 3, 32, 32), (3, 1, 5, 7), (2, 4, 256, 256)])
def test_fused_add_noise_is_bitwise_add_noise_of_the_noise_it_writes(shape):
    sch = d.DDPMScheduler()
    x0 = torch.from_numpy(synth.synth_scene_rasters(shape[0], shape[1], shape[2], shape[3], 5)).to(DEV)
    t = torch.tensor([0, 999, 417, 250][:shape[0]], device=DEV)
    noisy, noise = sch.add_noise_device(x0, t, seed=14555, offset=9)
    assert torch.equal(noise, ops.philox_normal(shape, 14555, 9))                 # the same stream as the plain fill
    assert torch.equal(noisy, sch.add_noise(x0, noise, t))                        # the library's own add_noise, bitwise
    # the tables the reference's way -- `alphas_cumprod ** 0.5` with torch on the host (diffusers' add_noise; on some CPUs torch's
    # vectorised pow is 1 ulp off numpy's sqrt, e.g. at t = 999 on the GPU box: the product follows the reference, so does this)
    ac = sch.alphas_cumprod
    sa, sb = (ac ** 0.5)[t.cpu()].numpy(), ((1 - ac) ** 0.5)[t.cpu()].numpy()
    want = po.add_noise(x0.cpu().numpy(), sa, sb, noise.cpu().numpy())
    assert np.array_equal(noisy.cpu().numpy(), want)                               # and the oracle's two-multiply-one-add
    again, nz2 = sch.add_noise_device(x0, t, seed=14555, offset=9)
    assert torch.equal(again, noisy) and torch.equal(nz2, noise)                   # stateless: (seed, offset) names the tensor
    other, nz3 = sch.add_noise_device(x0, t, seed=14555, offset=10)
    assert float((nz3 - noise).abs().max()) > 1.0
    with pytest.raises(RuntimeError, match="HIP engine"):
        sch.add_noise_device(x0.cpu(), t, 0, 0)


def test_train_steps_with_device_noise_is_reproducible_and_rank_disjoint():
    """`train_steps(..., noise="device")` on configs[0]'s network: two runs from the same weights, seeds and device-RNG state give
    the same losses bit for bit; another rank's stream gives different ones; the default ("host") still draws from the global
    CPU generator (its state moves), the device mode does not touch it."""
    from drivescenegen_amd import train_loop
    from tests.common import CFG1, synth_weights

    def run(noise):
        net = synth_weights(d.UNet2DModel(**CFG1)).to(DEV).train()      # (torch's default init draws from the global generator)
        opt = d.AdamW(net.parameters(), lr=1e-3)
        lrs = d.get_cosine_schedule_with_warmup(opt, 2, 10)
        acc = d.Accelerator()
        x0 = torch.from_numpy(synth.synth_scene_rasters(2, 3, 64, 64, 40)).to(DEV)
        torch.cuda.manual_seed(7)          # (the timesteps are the reference's device draw, training_pipeline.py:76)
        state = torch.get_rng_state()
        losses = [float(x) for x in train_loop.train_steps(acc, net, d.DDPMScheduler(), opt, lrs, [x0] * 3, noise=noise)]
        return losses, torch.equal(torch.get_rng_state(), state)
    a, quiet_a = run(train_loop.DeviceNoise(seed=1, rank=0))
    b, _ = run(train_loop.DeviceNoise(seed=1, rank=0))
    c, _ = run(train_loop.DeviceNoise(seed=1, rank=1))
    e, quiet_e = run("device")
    assert quiet_a and quiet_e             # the global CPU generator is never used by the loop in device mode
    assert a == b and a != c and all(np.isfinite(a)) and all(np.isfinite(c)) and all(np.isfinite(e))
    _, quiet_host = run("host")
    assert not quiet_host                  # the default draws the reference's host noise from it
    with pytest.raises(ValueError, match="noise must be"):
        list(train_loop.train_steps(None, None, None, None, None, [], noise="philox"))
