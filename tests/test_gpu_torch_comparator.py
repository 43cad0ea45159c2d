"""Independent full-size comparator (SURVEY App. D allows MIOpen / hipBLASLt for TEST cross-checks):
the oracle module itself (oracle/unet_oracle.py, the torch restatement of SURVEY App. A) is run by torch-ROCm on the
GPU -- MIOpen convolutions, rocBLAS / hipBLASLt GEMMs, torch's own GroupNorm and SDPA kernels: none of the engine's
code -- in fp32, and compared with the engine on EVERY row of configs[1] (batch 16) and on the 6-level 512x512
configs[3] network.  The CPU oracle stays the authority (it is what the other tests use); this one removes "one row
was spot-checked" and would expose a torch-CPU kernel quirk on the oracle side.  Never imported by the product."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import drivescenegen_amd as d  # noqa: E402
from oracle.unet_oracle import OracleUNet2DModel  # noqa: E402
from tests.common import CFG2, CFG4, max_abs, noisy_inputs, rel_l2, synth_weights  # noqa: E402

DEV = "cuda"


def _torch_gpu_reference(cfg, x, t, chunk):
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    ora = synth_weights(OracleUNet2DModel(**cfg)).eval().to(DEV)
    outs = []
    with torch.no_grad():
        for i in range(0, x.shape[0], chunk):
            outs.append(ora(x[i:i + chunk].to(DEV), t[i:i + chunk].to(DEV)).sample.cpu())
    return torch.cat(outs)


@pytest.mark.parametrize("name,cfg,batch,chunk", [("cfg2_b16", CFG2, 16, 4), ("cfg4_512_b2", CFG4, 2, 1)])
def test_engine_vs_torch_rocm_oracle_every_row(name, cfg, batch, chunk):
    x = noisy_inputs(cfg, batch)
    t = torch.arange(batch) * (980 // max(1, batch - 1))
    want = _torch_gpu_reference(cfg, x, t, chunk)
    net = synth_weights(d.UNet2DModel(**cfg)).to(DEV).eval().requires_grad_(False)
    got = net(x.to(DEV), t.to(DEV)).sample.cpu()
    assert torch.isfinite(got).all() and torch.isfinite(want).all()
    for i in range(batch):
        assert rel_l2(got[i], want[i]) <= 1e-4, (i, rel_l2(got[i], want[i]))
        assert max_abs(got[i], want[i]) <= 2e-4 * max(1.0, float(want[i].abs().max())), i
