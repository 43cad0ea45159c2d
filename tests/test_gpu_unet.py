"""Whole-network and sampler parity of the HIP engine against the CPU oracle and the golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import synth  # noqa: E402
from oracle.scheduler_oracle import OracleDDIMScheduler, OracleDDPMScheduler, oracle_pipeline  # noqa: E402
from oracle.unet_oracle import OracleUNet2DModel  # noqa: E402
from tests.common import (CFG1, CFG2, CFG4_SMALL, DEFAULT3, assert_matches_fullsize_golden, fullsize_case, max_abs,  # noqa: E402
                          noisy_inputs, rel_l2, synth_weights)

DEV = "cuda"
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg1_golden.npz"))
# fp32 single-forward tolerance
TOL_REL, TOL_ABS = 1e-4, 2e-4


def _pair(cfg):
    net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).eval().requires_grad_(False)
    ora = synth_weights(OracleUNet2DModel(**cfg)).eval()
    return net, ora


def _assert_close(got, want, rel=TOL_REL, ab=TOL_ABS):
    got = got.cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, want) <= rel, rel_l2(got, want)
    assert max_abs(got, want) <= ab * max(1.0, float(want.abs().max())), max_abs(got, want)


def _assert_trajectory(img, key):
    """Free-running 10-step trajectory (errors are amplified by up to 1/sqrt(abar_t) ~ 100 per step):
    relative L2 <= 1e-3 against the fp32 oracle, and the deviation from the fp64 ground truth
    is at most 3x the fp32 CPU oracle's own deviation from it."""
    got = img.cpu()
    f32, f64 = torch.from_numpy(GOLD[key + "_final"]), torch.from_numpy(GOLD[key + "_final_f64"])
    assert torch.isfinite(got).all()
    assert rel_l2(got, f32) <= 1e-3, rel_l2(got, f32)
    e_gpu, e_cpu = max_abs(got, f64), max_abs(f32, f64)
    assert e_gpu <= 3 * e_cpu + 1e-5, (e_gpu, e_cpu)
    r_gpu, r_cpu = rel_l2(got, f64), rel_l2(f32, f64)
    assert r_gpu <= 3 * r_cpu + 1e-6, (r_gpu, r_cpu)


@pytest.fixture(scope="module")
def tiny():
    return _pair(CFG1)


def test_cfg1_forward_vs_golden_and_oracle(tiny):
    net, ora = tiny
    x = noisy_inputs(CFG1, 2)
    assert abs(float(np.abs(x.numpy()).sum()) - float(GOLD["input_checksum"][0])) < 1e-3
    for t in (0, 1, 499, 999):
        got = net(x.to(DEV), t).sample
        _assert_close(got, torch.from_numpy(GOLD[f"fwd_t{t}"]))
        with torch.no_grad():
            _assert_close(got, ora(x, t).sample)
    tv = torch.tensor([17, 801])
    got = net(x.to(DEV), tv.to(DEV), return_dict=False)[0]
    _assert_close(got, torch.from_numpy(GOLD["fwd_tvec"]))
    # 0-d tensor timestep, as the pipeline loop passes it
    got = net(x.to(DEV), torch.tensor(499)).sample
    _assert_close(got, torch.from_numpy(GOLD["fwd_t499"]))


def test_cfg1_trajectories_vs_golden(tiny):
    net, _ = tiny
    sch = d.DDPMScheduler()
    sch.set_timesteps(10)
    img = torch.from_numpy(synth.normal(9, (2, 3, 64, 64))).to(DEV)
    for i, t in enumerate(sch.timesteps):
        eps = net(img, t).sample
        z = torch.from_numpy(synth.normal(9, (2, 3, 64, 64), stream=100 + i)).to(DEV) if int(t) > 0 else None
        img = sch.step(eps, t, img, variance_noise=z).prev_sample
    _assert_trajectory(img, "ddpm10")
    dd = d.DDIMScheduler()
    dd.set_timesteps(10)
    img = torch.from_numpy(synth.normal(9, (2, 3, 64, 64))).to(DEV)
    for t in dd.timesteps:
        img = dd.step(net(img, t).sample, t, img).prev_sample
    _assert_trajectory(img, "ddim10")


def test_cfg1_pipeline_seeded_generator_matches_oracle(tiny):
    """training_pipeline.py:26-32: CPU generator seed -> device-independent noise stream (App. A.4)."""
    net, ora = tiny
    pipe = d.DDPMPipeline(unet=net, scheduler=d.DDPMScheduler())
    got = pipe(num_inference_steps=10, batch_size=2, generator=torch.manual_seed(14555), output_type="np.array",
               return_dict=False)[0]
    want = oracle_pipeline(ora, OracleDDPMScheduler(), batch_size=2, generator=torch.manual_seed(14555),
                           num_inference_steps=10, output_type="np.array")
    assert got.shape == (2, 64, 64, 3) and got.dtype == np.float32
    assert rel_l2(torch.from_numpy(got), torch.from_numpy(want)) <= 1e-3
    assert float(np.abs(got - want).max()) <= 1e-2
    # sharded sampling (no collective) reproduces the rows of the unsharded run
    parts = [pipe(num_inference_steps=10, batch_size=2, generator=torch.manual_seed(14555), output_type="np.array",
                  return_dict=False, shard=(r, 2))[0] for r in range(2)]
    assert np.array_equal(np.concatenate(parts), got)  # bitwise: batch rows are independent
    # uint8 outputs: <= 1 LSB, on a small fraction of pixels (values that sit on a rounding boundary)
    pil = pipe(num_inference_steps=10, batch_size=2, generator=torch.manual_seed(14555)).images
    w8 = (want * 255).round().astype("uint8")
    g8 = np.stack([np.asarray(im) for im in pil])
    diff = np.abs(g8.astype(int) - w8.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() <= 5e-3, (diff.max(), (diff > 0).mean())


def test_pipeline_noise_stream_is_bitwise_the_reference_order_of_draws(tiny):
    """DDPMPipeline draws a CPU generator's noise into a ring of three pinned buffers one step ahead of its use, and the step
    kernel reads the buffer in place over PCIe at the device address the runtime reports for it (pipelines._NoiseStream,
    schedulers.HostNoise -- no copy, no side stream); the U-Net gets its timesteps from a device table uploaded once.  Same generator, same shapes,
    same order as diffusers' loop (randn_tensor(...).to(device) per step, a host scalar timestep per step): the images must
    be those of the plain loop, bit for bit -- also for a shard, also when the call is repeated on the same pipeline."""
    from drivescenegen_amd.schedulers import _randn_like_reference
    net, _ = tiny
    sch = d.DDPMScheduler()
    pipe = d.DDPMPipeline(unet=net, scheduler=sch)
    n = 25

    def plain(batch, rows=None):
        gen = torch.manual_seed(4242)
        x = _randn_like_reference((batch, 3, 64, 64), gen, DEV, torch.float32)
        x = x if rows is None else x[rows].contiguous()
        sch.set_timesteps(n)
        for t in sch.timesteps:
            eps = net(x, t).sample
            z = None
            if int(t) > 0:
                z = _randn_like_reference((batch, 3, 64, 64), gen, DEV, torch.float32)
                z = z if rows is None else z[rows].contiguous()
            x = sch.step(eps, t, x, variance_noise=z).prev_sample
        return (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy()

    want = plain(3)
    for _ in range(2):
        got = pipe(num_inference_steps=n, batch_size=3, generator=torch.manual_seed(4242), output_type="np.array",
                   return_dict=False)[0]
        assert np.array_equal(got, want)
    part = pipe(num_inference_steps=n, batch_size=4, generator=torch.manual_seed(4242), output_type="np.array",
                return_dict=False, shard=(1, 2))[0]
    assert np.array_equal(part, plain(4, slice(2, 4)))
    # a call that dies mid-loop leaves the generator where the serial loop would have (the worker's extra draw is undone)
    calls = {"n": 0}
    real_step = sch.step

    def failing_step(*a, **k):
        calls["n"] += 1
        if calls["n"] == 4:
            raise KeyError("step 4 failed")
        return real_step(*a, **k)
    sch.step = failing_step
    # (a PRIVATE generator: torch.manual_seed returns the global one, and comparing the global generator with itself proves
    #  nothing -- ADVICE r05)
    gen = torch.Generator().manual_seed(4242)
    with pytest.raises(KeyError):
        pipe(num_inference_steps=n, batch_size=3, generator=gen, output_type="np.array")
    sch.step = real_step
    state_after_failure = gen.get_state().clone()
    ref = torch.Generator().manual_seed(4242)
    for _ in range(5):                                # x_T + the draws of steps 1..4 (the one that died drew its noise before failing)
        torch.randn((3, 3, 64, 64), generator=ref)
    assert torch.equal(state_after_failure, ref.get_state())
    one_more = torch.Generator().manual_seed(4242)
    for _ in range(6):
        torch.randn((3, 3, 64, 64), generator=one_more)
    assert not torch.equal(state_after_failure, one_more.get_state())     # (the worker's extra draw really was undone)
    # no generator: torch's device RNG, as the reference (generation.py:14-20) -- reproducible under torch.cuda.manual_seed
    torch.cuda.manual_seed(99)
    a = pipe(num_inference_steps=5, batch_size=2, output_type="np.array").images
    torch.cuda.manual_seed(99)
    b = pipe(num_inference_steps=5, batch_size=2, output_type="np.array").images
    assert np.array_equal(a, b) and np.isfinite(a).all()


def test_cfg1_ddim_pipeline_matches_oracle(tiny):
    net, ora = tiny
    pipe = d.DDIMPipeline(unet=net, scheduler=d.DDIMScheduler())
    got = pipe(num_inference_steps=10, batch_size=2, generator=torch.manual_seed(7), output_type="np.array").images
    want = oracle_pipeline(ora, OracleDDIMScheduler(), batch_size=2, generator=torch.manual_seed(7),
                           num_inference_steps=10, output_type="np.array", ddim=True)
    assert rel_l2(torch.from_numpy(got), torch.from_numpy(want)) <= 1e-3
    assert float(np.abs(got - want).max()) <= 1e-2


def test_attn_blocks_forward_vs_oracle():
    """AttnDownBlock2D / AttnUpBlock2D (BASELINE configs[3]'s block types), spatially shrunk."""
    net, ora = _pair(CFG4_SMALL)
    x = noisy_inputs(CFG4_SMALL, 2)
    t = torch.tensor([990, 10])
    with torch.no_grad():
        want = ora(x, t).sample
    _assert_close(net(x.to(DEV), t.to(DEV)).sample, want)


@pytest.mark.parametrize("cfg", [CFG1, CFG4_SMALL], ids=["cfg1_tiny", "cfg4_attention_small"])
def test_blocked_and_plain_intermediates_agree(cfg):
    """dsg_unet_forward keeps its intermediates channel-blocked ([N,C/8,H,W,8], dsg_set_tuning key 13); the same
    plan run with [N,C,H,W] intermediates gives the same result to round-off (same kernels' arithmetic; only the
    summation order of the fallback GroupNorm statistics differs)."""
    from drivescenegen_amd import _lib
    net, _ = _pair(cfg)
    x = noisy_inputs(cfg, 2).to(DEV)
    lib = _lib.load()
    try:
        with torch.no_grad():
            _lib.check(lib.dsg_set_tuning(13, 0))
            plain = net(x, 321).sample.clone()
            _lib.check(lib.dsg_set_tuning(13, 1))
            blocked = net(x, 321).sample.clone()
    finally:
        lib.dsg_set_tuning(13, 1)
    assert rel_l2(blocked.cpu(), plain.cpu()) <= 2e-6


def test_default_unet_256_forward_vs_oracle():
    """The train.py:39-57 network itself (56,574,595 params), B=1, 256x256x3, against the oracle's stored output
    (tests/golden/fullsize_golden.npz; tests/test_oracle_kat.py re-runs the oracle against the same entry on the CPU)."""
    net = synth_weights(d.UNet2DModel(**DEFAULT3)).to(DEV).eval().requires_grad_(False)
    assert sum(p.numel() for p in net.parameters()) == 56_574_595
    _, _, x, t, _ = fullsize_case("default3_b1_t749")
    assert torch.equal(x, noisy_inputs(DEFAULT3, 1)) and t == 749
    assert_matches_fullsize_golden(net(x.to(DEV), t).sample, "default3_b1_t749", TOL_REL, TOL_ABS)


def test_cfg2_full_size_batch_properties():
    """BASELINE configs[1] shape (256x256x4, B=16): size-independent properties instead of a CPU oracle run --
    batch rows are independent (row i of a B=16 call == the B=1 call on row i, bitwise: the kernels'
    reduction order does not depend on batch) and row 0 against the oracle's stored output."""
    net = synth_weights(d.UNet2DModel(**CFG2)).to(DEV).eval().requires_grad_(False)
    x = noisy_inputs(CFG2, 16)
    t = torch.full((16,), 500, dtype=torch.long)
    t[1] = 20
    out = net(x.to(DEV), t.to(DEV)).sample
    assert torch.isfinite(out).all()
    from tests.common import same_kernels_at_any_batch
    for i in (0, 1, 15):
        with same_kernels_at_any_batch():
            one = net(x[i:i + 1].to(DEV), t[i:i + 1].to(DEV)).sample
        assert torch.equal(one[0], out[i]), i
        # the default batch-1 path (split-K on the deep, small-grid layers): the same values to fp32 round-off
        fast = net(x[i:i + 1].to(DEV), t[i:i + 1].to(DEV)).sample
        assert rel_l2(fast[0].cpu(), out[i].cpu()) <= 2e-6, rel_l2(fast[0].cpu(), out[i].cpu())
    assert torch.equal(fullsize_case("cfg2_b16_row0_t500")[2], x[:1]) and int(t[0]) == 500
    assert_matches_fullsize_golden(out[:1], "cfg2_b16_row0_t500", TOL_REL, TOL_ABS)


def test_checkpoint_roundtrip(tmp_path, tiny):
    """save_pretrained / from_pretrained keep the App. A.5 folder layout (training_pipeline.py:107, generation.py:7)."""
    net, _ = tiny
    pipe = d.DDPMPipeline(unet=net, scheduler=d.DDPMScheduler())
    pipe.save_pretrained(str(tmp_path))
    for rel in ("model_index.json", "unet/config.json", "unet/diffusion_pytorch_model.bin",
                "scheduler/scheduler_config.json"):
        assert (tmp_path / rel).exists(), rel
    re = d.DDPMPipeline.from_pretrained(str(tmp_path), variant="fp16").to(DEV)
    re.unet.requires_grad_(False)
    x = noisy_inputs(CFG1, 1).to(DEV)
    assert torch.equal(re.unet(x, 3).sample, net(x, 3).sample)
    u2 = d.UNet2DModel.from_pretrained(str(tmp_path), subfolder="unet")
    assert list(u2.state_dict().keys()) == list(net.state_dict().keys())


def test_param_update_is_picked_up(tiny):
    net, _ = tiny
    x = noisy_inputs(CFG1, 1).to(DEV)
    a = net(x, 5).sample
    with torch.no_grad():
        net.conv_out.bias.add_(1.0)
    b = net(x, 5).sample
    assert torch.allclose(b - a, torch.ones_like(a), atol=1e-5)
    with torch.no_grad():
        net.conv_out.bias.sub_(1.0)


@pytest.mark.parametrize("mode", ["eval", "train_fp32", "train_bf16"])
def test_wrong_channel_count_is_refused_before_any_kernel(mode):
    """A sample whose channel count is not config.in_channels raises in forward() -- inference plan and both training tapes --
    instead of handing the conv_in kernel a tensor it indexes past the end of (diffusers' UNet2DModel fails in F.conv2d with a
    shape error; a probe script that passed 4 channels to the 8-channel network ended in a GPU memory fault before this check)."""
    net = synth_weights(d.UNet2DModel(**CFG1)).to(DEV)
    if mode == "eval":
        net.eval().requires_grad_(False)
    else:
        net.train().set_compute_dtype("fp32" if mode == "train_fp32" else "bf16")
    bad = torch.zeros(2, CFG1["in_channels"] + 1, CFG1["sample_size"], CFG1["sample_size"], device=DEV)
    with pytest.raises(ValueError):
        net(bad, 5)
    with pytest.raises(ValueError):
        net(bad[:, :1], 5)
    with pytest.raises(ValueError):
        net(bad[0, :CFG1["in_channels"]], 5)
    odd = torch.zeros(2, CFG1["in_channels"], CFG1["sample_size"] + 1, CFG1["sample_size"], device=DEV)   # 65 rows: the skip maps would not match
    with pytest.raises((ValueError, RuntimeError)):
        net(odd, 5)


def test_batch_invariant_flag_and_tuning_gate():
    """dsg_unet_config.flags / DSG_UNET_BATCH_INVARIANT (UNet2DModel.batch_invariant): per-plan, no process-global state --
    row i of a batch is bitwise the batch-1 call; the default plan (split-K on small grids) agrees to fp32 round-off.
    And the process-global switches are a test hook: without DSG_TESTING=1 in the environment dsg_set_tuning refuses."""
    import subprocess
    import sys
    net = synth_weights(d.UNet2DModel(**CFG4_SMALL)).to(DEV).eval().requires_grad_(False)
    x = noisy_inputs(CFG4_SMALL, 4).to(DEV)
    t = torch.tensor([900, 500, 100, 3], device=DEV)
    fast = net(x, t).sample.clone()
    net.batch_invariant = True
    full = net(x, t).sample.clone()
    for i in range(4):
        assert torch.equal(net(x[i:i + 1], t[i:i + 1]).sample[0], full[i]), i
    assert rel_l2(fast.cpu(), full.cpu()) <= 2e-6
    code = ("import os; os.environ.pop('DSG_TESTING', None); os.environ.pop('DSG_TUNING', None)\n"
            "from drivescenegen_amd import _lib\n"
            "lib = _lib.load(); rc = lib.dsg_set_tuning(19, 0); print('rc', rc, lib.dsg_last_error().decode()[:120])")
    env = {k: v for k, v in os.environ.items() if k not in ("DSG_TESTING", "DSG_TUNING")}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "rc -1" in out.stdout and "test hook" in out.stdout, out.stdout


def test_forward_and_scheduler_step_are_legal_under_stream_capture():
    """include/dsg.h's contract: every entry point is stream-asynchronous.  Until round 4 `dsg_unet_set_param` synchronised the
    stream per conv weight (illegal inside a capture); now a parameter refresh is asynchronous + ONE `dsg_unet_commit_params`.
    Here a denoising step -- dsg_unet_forward + dsg_ddim_step -- is captured into a HIP graph after one eager warm-up (which
    also settles the plan's parameters), replayed on fresh inputs, and must reproduce the eager results bit for bit."""
    net = synth_weights(d.UNet2DModel(**CFG1)).to(DEV).eval().requires_grad_(False)
    sch = d.DDIMScheduler()
    sch.set_timesteps(50)
    t = int(sch.timesteps[3])
    x_static = noisy_inputs(CFG1, 2).to(DEV)
    t_dev = torch.full((2,), t, dtype=torch.long, device=DEV)   # (a host scalar would be an H2D copy inside the capture)
    eager_eps = net(x_static, t_dev).sample
    eager_prev = sch.step(eager_eps, t, x_static).prev_sample
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):      # (torch's capture recipe: warm up on the side stream first)
        net(x_static, t_dev)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eps = net(x_static, t_dev).sample
        prev = sch.step(eps, t, x_static).prev_sample
    for _ in range(3):   # (replays 2.. are the ones that found the un-ordered memset node: dsg::zero_words)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(eps, eager_eps) and torch.equal(prev, eager_prev)
    for seed in (77, 78, 79):
        x2 = noisy_inputs(CFG1, 2, seed=seed).to(DEV)
        want_eps = net(x2, t_dev).sample
        want_prev = sch.step(want_eps, t, x2).prev_sample
        x_static.copy_(x2)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(eps, want_eps) and torch.equal(prev, want_prev), seed
