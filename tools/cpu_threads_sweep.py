"""Pick a sane thread count for the CPU-oracle baseline on the GPU box (256 logical CPUs oversubscribe torch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.unet_oracle import OracleUNet2DModel
from tests.common import CFG2, synth_weights
net = synth_weights(OracleUNet2DModel(**CFG2)).eval()
x = torch.randn(2, 4, 256, 256)
for th in [int(a) for a in sys.argv[1:]]:
    torch.set_num_threads(th)
    with torch.no_grad():
        net(x, 10)
        t0 = time.perf_counter(); net(x, 10); dt = time.perf_counter() - t0
    print(f"threads {th}: {dt:.2f} s per B=2 forward -> {2/dt:.3f} image-steps/s", flush=True)
