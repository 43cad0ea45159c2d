"""dsg_conv_weight_pack_batch: every conv weight's operand images (and the time-embedding projection rows) refreshed in ONE launch
after an optimizer step (the reference's loop, training_pipeline.py:84-91: optimizer.step() at :89 changes every weight; the
mixed-precision tape used to re-pack them in ~190 launches).  Same bits as the one-by-one dsg_conv_weight_pack calls, for every
kind / dtype / column window, and a training run that uses it is bitwise the run that does not."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from drivescenegen_amd import ops, synth  # noqa: E402
from tests.common import CFG1, CFG4_SMALL, synth_weights  # noqa: E402

DEV = "cuda"


def _t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * float(scale)).astype(np.float32)).to(DEV)


def test_pack_batch_matches_one_by_one():
    jobs, want = [], []
    seed = 0
    for dt in ("fp32", "bf16", "fp16"):
        for (cout, cin, k, kind) in [(64, 32, 3, ops.PACK_FWD), (96, 64, 3, ops.PACK_FWD), (128, 64, 3, ops.PACK_DGRAD),
                                     (64, 64, 3, ops.PACK_FOLD), (64, 32, 3, ops.PACK_S2), (64, 48, 3, ops.PACK_DGRAD_S2), (40, 72, 3, ops.PACK_DGRAD_UPS),
                                     (128, 96, 1, ops.PACK_FWD), (64, 128, 1, ops.PACK_DGRAD), (8, 64, 3, ops.PACK_FWD)]:
            seed += 1
            w = _t(seed, (cout, cin, k, k) if k == 3 else (cout, cin), 0.1)
            ref = ops.pack_conv_weight(w, kind, dt)
            dst = torch.zeros_like(ref)
            jobs.append(dict(w=w, dst=dst, kind=kind, dtype=dt))
            want.append(ref)
        # three column windows of one wide matrix (the fused q / k / v projection)
        c = 64
        ws = [_t(100 + seed + i, (c, c), 0.2) for i in range(3)]
        ref = None
        for i, w in enumerate(ws):
            ref = ops.pack_conv_weight(w, ops.PACK_FWD, dt, n_total=3 * c, n_off=i * c, out=ref)
        dst = torch.zeros_like(ref)
        for i, w in enumerate(ws):
            jobs.append(dict(w=w, dst=dst, kind=ops.PACK_FWD, dtype=dt, n_total=3 * c, n_off=i * c))
        want.append(ref)
    # plain copies (rows of the fused time_emb_proj matrix)
    big = torch.zeros(7, 33, device=DEV)
    src = _t(999, (3, 33))
    jobs.append(dict(w=src, dst=big[2:5], kind=-1))
    table = ops.PackTable(jobs)
    table.run()
    torch.cuda.synchronize()
    outs, seen = [], set()
    for j in jobs[:-1]:
        if j["dst"].data_ptr() not in seen:
            seen.add(j["dst"].data_ptr())
            outs.append(j["dst"])
    assert len(outs) == len(want)
    for got, ref in zip(outs, want):
        assert torch.equal(got, ref)
    assert torch.equal(big[2:5], src) and float(big[:2].abs().max()) == 0.0 and float(big[5:].abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [CFG1, CFG4_SMALL], ids=["cfg1_tiny", "attn_blocks"])
def test_training_with_batched_refresh_is_bitwise_the_one_by_one_run(cfg):
    def run(batch):
        if batch:
            os.environ.pop("DSG_NO_PACK_BATCH", None)
        else:
            os.environ["DSG_NO_PACK_BATCH"] = "1"
        try:
            net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).train().set_compute_dtype("bf16")
            opt = d.AdamW(net.parameters(), lr=1e-3)
            ss = cfg["sample_size"]
            x0 = torch.from_numpy(synth.synth_scene_rasters(2, cfg["in_channels"], ss, ss, 5)).to(DEV)
            noise = torch.from_numpy(synth.normal(6, tuple(x0.shape))).to(DEV)
            t = torch.tensor([12, 900], device=DEV)
            noisy = d.DDPMScheduler().add_noise(x0, noise, t)
            losses = []
            for _ in range(4):
                loss = d.mse_loss(net(noisy, t, return_dict=False)[0], noise)
                loss.backward()
                opt.step()
                opt.zero_grad()
                losses.append(float(loss.detach().cpu()))
            return losses, [p.detach().clone() for p in net.parameters()]
        finally:
            os.environ.pop("DSG_NO_PACK_BATCH", None)
    la, pa = run(True)
    lb, pb = run(False)
    assert la == lb, (la, lb)
    assert la[-1] < la[0]   # (it trains)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
