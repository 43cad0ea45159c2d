"""Winograd F(2x2, 3x3) for the deep fp32-equivalent convs: is there a 1.3x to be had?

F(2x2,3x3) replaces the 9 tap products of a 2x2 output tile by 16 element-wise products in the transform domain: 2.25x fewer
MFMAs.  On the fp16x2-split path the transformed operands V = B^T d B (input tiles) and U = G g G^T (weights) are fp32
quantities that must be split into (hi, scaled lo) fp16 pairs like any other operand, i.e. 4 bytes per transformed value, and
there are 16 transformed values per 4 input pixels: the transform domain is 4x the tensor.

(a) Fused in one kernel (transform in the staging pass, 16 per-position GEMMs on the accumulators, inverse transform in the
    epilogue): a workgroup needs 16 positions x couts x tiles x (hi, lo) accumulators.  The register file that holds
    64 couts x 512 pixels of direct-conv accumulators holds 64 couts x 32 tiles (= 128 pixels) of Winograd ones, and the
    weight slab a K-chunk needs is 16 positions x 16 channels x 64 couts x 4 B = 64 KB for 96 MFMAs per workgroup -- 682 B of
    L2 -> LDS weight traffic per MFMA against 42 B in the direct kernel (85 B/clk/CU against a 64 B/clk/CU path).  Not
    realisable on this register file / LDS.
(b) Unfused (input-transform pass -> 16 batched GEMMs -> output-transform pass): the GEMM stage is what this script times, by
    proxy: a 1x1 conv Cin -> Cout over 4x the pixels (the same MACs, the same 4 B/element operand and result streams as the 16
    position-GEMMs over N*H*W/4 tiles) on the library's own split 1x1 kernel.  The two transform passes are priced at the
    streaming bandwidth measured next to it (read X + write V = 5x the tensor; read M + write Y (+ residual) = 5-6x).

Prints the direct conv's time, the GEMM-stage proxy, the transform passes at measured bandwidth, and the ratio.
Usage: python tools/winograd_bound.py      (one MI355X; numbers for the 32^2 x 512 and 64^2 x 256 layers at batch 16)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops

B = 16
dev = "cuda"


def timeit(f, iters=30):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


# streaming bandwidth of this box on a transform-domain-sized tensor (134 MB): read + write
big = torch.randn(16 * 512 * 64 * 64, device=dev)
out = torch.empty_like(big)
us = timeit(lambda: out.copy_(big))
bw = 2 * big.numel() * 4 / us / 1e6   # TB/s
print(f"streaming copy of {big.numel() * 4 / 1e6:.0f} MB: {us:.1f} us = {bw:.2f} TB/s (read + write)")
del big, out

for c, h in ((512, 32), (256, 64)):
    x = ops.to_blocked(torch.randn(B, c, h, h, device=dev))
    r = ops.to_blocked(torch.randn(B, c, h, h, device=dev))
    ss = torch.randn(B, c, 2, device=dev)
    w3 = torch.randn(c, c, 3, 3, device=dev) * 0.02
    bias = torch.randn(c, device=dev)
    wh3 = ops.relayout_conv_weight_h2(w3)
    direct = timeit(lambda: ops.conv2d_fused(x, None, bias, ksize=3, cout=c, gn_scale_shift=ss, silu=True, residual=r,
                                             weight_h2=wh3, src_blocked=True, dst_blocked=True))
    # the 16 position-GEMMs [Cout x Cin] x [Cin x N*H*W/4] == one 1x1 conv over 4x the pixels (2H x 2W)
    x4 = ops.to_blocked(torch.randn(B, c, 2 * h, 2 * h, device=dev))
    w1 = torch.randn(c, c, 1, 1, device=dev) * 0.02
    wh1 = ops.relayout_conv_weight_h2(w1)
    gemm = timeit(lambda: ops.conv2d_fused(x4, None, None, ksize=1, cout=c, weight_h2=wh1, src_blocked=True, dst_blocked=True))
    t_bytes = B * c * h * h * 4
    tin = (t_bytes + 4 * t_bytes) / bw / 1e6          # read X, write V (us)
    tout = (4 * t_bytes + 2 * t_bytes) / bw / 1e6     # read M, read residual, write Y
    fl = 2.0 * B * h * h * c * c * 9
    print(f"{c}->{c} @ {h}^2, batch {B}: direct 3x3 (GN + SiLU + residual fused) {direct:7.1f} us = {fl / direct / 1e6:6.1f} TF/s-eq | "
          f"Winograd stages: GEMM proxy {gemm:7.1f} us + input transform >= {tin:5.1f} us + output transform >= {tout:5.1f} us "
          f"= {gemm + tin + tout:7.1f} us -> {direct / (gemm + tin + tout):.2f}x (GEMM stage alone {direct / gemm:.2f}x)")
