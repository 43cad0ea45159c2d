"""Chunk-end wait breakdown of conv_h2 (needs lib built with tools/build_variant.sh timing -DDSG_H2_TIMING,
run with DSG_LIB_PATH=drivescenegen_amd/lib/libdsg_timing.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops
for name, c, h, B in (("res512@32", 512, 32, 16), ("res256@64", 256, 64, 16), ("res128@128", 128, 128, 16),
                      ("res64@256", 64, 256, 16), ("res64@64 B=256", 64, 64, 256), ("res64@512 B=4", 64, 512, 4)):
    blk = os.environ.get("BLOCKED") == "1"
    x = torch.randn(B, c, h, h, device="cuda")
    if blk:
        x = ops.to_blocked(x)
    w = torch.randn(c, c, 3, 3, device="cuda") * 0.05
    wr, wh = ops.relayout_conv_weight(w), ops.relayout_conv_weight_h2(w)
    ss = torch.randn(B, c, 2, device="cuda")
    _, st = ops.conv2d_fused(x, wr, None, gn_scale_shift=ss, silu=True, weight_h2=wh, want_stats=True, src_blocked=blk, dst_blocked=blk)
    torch.cuda.synchronize()
    st.zero_()
    ops.conv2d_fused(x, wr, None, gn_scale_shift=ss, silu=True, weight_h2=wh, want_stats=True, stats_buf=st, src_blocked=blk, dst_blocked=blk)
    torch.cuda.synchronize()
    tot, vm, bar, nw = st.flatten()[:4].tolist()
    print(f"{name}: waves {nw:.0f}  cycles/wave {tot / nw:.0f}  wait-vmcnt {100 * vm / tot:.1f}%  barrier {100 * bar / tot:.1f}%  "
          f"per chunk: total {tot / nw / (c // 16):.0f} vm {vm / nw / (c // 16):.0f} bar {bar / nw / (c // 16):.0f}")
    import numpy as np
    nb = int(nw) // 4
    rec = st.flatten()[8:8 + 4 * nb].reshape(nb, 4).cpu().numpy()
    end = st.flatten()[8 + 4 * nb:8 + 5 * nb].cpu().numpy()
    t0 = rec[:, 0].min()
    us = lambda a: (a - t0) / 100.0
    print(f"   blocks {nb}: start {us(rec[:,0]).min():.1f}..{us(rec[:,0]).max():.1f} us  loop-begin {us(rec[:,1]).min():.1f}..{us(rec[:,1]).max():.1f}"
          f"  loop-end {us(rec[:,2]).min():.1f}..{us(rec[:,2]).max():.1f}  block-end {us(end).min():.1f}..{us(end).max():.1f}"
          f"  loop cycles min/mean/max {rec[:,3].min():.0f}/{rec[:,3].mean():.0f}/{rec[:,3].max():.0f}"
          f"  PHASES us: prologue {(rec[:,1]-rec[:,0]).mean()/100:.2f} loop {(rec[:,2]-rec[:,1]).mean()/100:.2f} epilogue {(end-rec[:,2]).mean()/100:.2f}"
          f"  loop us mean {(rec[:,2]-rec[:,1]).mean()/100:.1f} -> {rec[:,3].mean()/((rec[:,2]-rec[:,1]).mean()/100)/1e3:.2f} GHz")
    taps = st.flatten()[8 + 5 * nb:8 + 5 * nb + 9].cpu().numpy() / nw / (c // 16)
    print("   cycles per tap:", " ".join(f"{t:.0f}" for t in taps), " sum", f"{taps.sum():.0f}")
    pr = st.flatten()[8 + 5 * nb + 16:8 + 5 * nb + 16 + 4 * nb].reshape(nb, 4).cpu().numpy()
    e = rec[:, 0]
    print("   prologue us: entry->p0 %.2f  issue loads %.2f  ss+sync (first wait) %.2f  commit %.2f  final wait+barrier %.2f" % (
        (pr[:, 0] - e).mean() / 100, (pr[:, 1] - pr[:, 0]).mean() / 100, (pr[:, 2] - pr[:, 1]).mean() / 100,
        (pr[:, 3] - pr[:, 2]).mean() / 100, (rec[:, 1] - pr[:, 3]).mean() / 100))
    er = st.flatten()[8 + 9 * nb + 16:8 + 9 * nb + 16 + 6 * nb].reshape(nb, 6).cpu().numpy()
    d = lambda a, b: (er[:, b] - er[:, a]).mean() / 100
    print("   epilogue us: slab0 loads land %.2f  slab0 math+stores %.2f  slab1 loads land %.2f  slab1 math+stores %.2f  stats tail %.2f" % (
        d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5)))
