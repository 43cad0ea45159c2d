"""Chunk-end wait breakdown of conv_h2 (needs lib built with tools/build_variant.sh timing -DDSG_H2_TIMING,
run with DSG_LIB_PATH=drivescenegen_amd/lib/libdsg_timing.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drivescenegen_amd import ops
for name, c, h, B in (("res512@32", 512, 32, 16), ("res256@64", 256, 64, 16), ("res128@128", 128, 128, 16),
                      ("res64@256", 64, 256, 16), ("res64@64 B=256", 64, 64, 256), ("res64@512 B=4", 64, 512, 4)):
    blk = os.environ.get("BLOCKED") == "1"
    dt = os.environ.get("DTYPE", "fp32")   # bf16 / fp16: the 16-bit kernels (needs BLOCKED=1 and a timing build of conv_h2_bf16.hip)
    if os.environ.get("BSCALE"):
        B = B * int(os.environ["BSCALE"])
    x = torch.randn(B, c, h, h, device="cuda")
    if blk:
        x = ops.to_blocked(x, dt) if dt != "fp32" else ops.to_blocked(x)
    w = torch.randn(c, c, 3, 3, device="cuda") * 0.05
    wr = ops.relayout_conv_weight(w)
    wh = ops.relayout_conv_weight_h2(w) if dt == "fp32" else ops.pack_conv_weight(w, ops.PACK_FWD, dt)
    ss = torch.randn(B, c, 2, device="cuda")
    kw = dict(gn_scale_shift=ss, silu=True, weight_h2=wh, want_stats=True, src_blocked=blk, dst_blocked=blk,
              compute_dtype=dt if dt != "fp32" else 0)
    _, st = ops.conv2d_fused(x, wr, None, **kw)
    torch.cuda.synchronize()
    st.zero_()
    ops.conv2d_fused(x, wr, None, stats_buf=st, **kw)
    torch.cuda.synchronize()
    import numpy as np
    # the grid is not known here exactly (8- or 16-row tiles, 64 or 128 couts): count the records that were written
    recs = st.flatten().cpu().numpy()
    recs = recs[:recs.size // 32 * 32].reshape(-1, 32)
    recs = recs[recs[:, 0] > 0]
    nb = len(recs)
    nq = c // 16
    cyc, vm, bar = recs[:, 3], recs[:, 4], recs[:, 5]
    t0 = recs[:, 0].min()
    us = lambda a: (a - t0) / 100.0
    loop_us = (recs[:, 2] - recs[:, 1]) / 100.0
    print(f"{name} [{dt}]: blocks {nb}  kernel {us(recs[:, 16]).max():.1f} us  cycles/chunk {cyc.mean() / nq:.0f} (vm wait {vm.mean() / nq:.0f}, barrier {bar.mean() / nq:.0f})"
          f"  PHASES us: prologue {(recs[:, 1] - recs[:, 0]).mean() / 100:.2f} loop {loop_us.mean():.2f} epilogue {(recs[:, 16] - recs[:, 2]).mean() / 100:.2f}"
          f"  clock {cyc.mean() / loop_us.mean() / 1e3:.2f} GHz")
    print("   cycles per tap:", " ".join(f"{t:.0f}" for t in recs[:, 17:26].mean(0) / nq), f" sum {recs[:, 17:26].mean(0).sum() / nq:.0f}")
    pr = recs[:, 6:10]
    print("   prologue us: entry->p0 %.2f  issue loads %.2f  ss+sync (first wait) %.2f  commit %.2f  final wait+barrier %.2f" % (
        (pr[:, 0] - recs[:, 0]).mean() / 100, (pr[:, 1] - pr[:, 0]).mean() / 100, (pr[:, 2] - pr[:, 1]).mean() / 100,
        (pr[:, 3] - pr[:, 2]).mean() / 100, (recs[:, 1] - pr[:, 3]).mean() / 100))
    er = recs[:, 10:16]
    d = lambda a, b: (er[:, b] - er[:, a]).mean() / 100
    print("   epilogue us: slab0 loads land %.2f  slab0 math+stores %.2f  slab1 loads land %.2f  slab1 math+stores %.2f  stats tail %.2f" % (
        d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5)))
    if os.environ.get("PERS") == "1":   # persistent variant: the tap slots carry (loop end, tile end) of the first four tiles
        pt = (recs[:, 17:25] - recs[:, 0:1]) / 100.0
        print("   persistent: entry->loop %.2f us; tile k (K loop done, tile done) since entry: " % ((recs[:, 1] - recs[:, 0]).mean() / 100)
              + "  ".join(f"({pt[:, 2 * i].mean():.1f}, {pt[:, 2 * i + 1].mean():.1f})" for i in range(4))
              + f"   workgroup total {((recs[:, 16] - recs[:, 0]) / 100).mean():.1f} us")
