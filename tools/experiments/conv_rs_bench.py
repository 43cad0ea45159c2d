"""One 64 -> 64 channel 3x3 conv (GroupNorm + SiLU in front, bias + temb + residual + statistics behind) on the row-streaming
kernel (csrc/conv_rs.hip, key 33) and on conv_h2_kernel; DSG_LIB_PATH selects an ablation build (tools/build_variant.sh
rs_X conv_rs.hip -DRS_ABL_{NOFIN,NOSTAGE,NOBAR,NOMMA}).   Usage: conv_rs_bench.py [h] [w] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DSG_TESTING", "1")
import numpy as np, torch
from drivescenegen_amd import ops, synth, _lib
h = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w = int(sys.argv[2]) if len(sys.argv) > 2 else 512
b = int(sys.argv[3]) if len(sys.argv) > 3 else 8
t = lambda s, shape, sc=1.0: torch.from_numpy((synth.normal(s, shape) * sc).astype(np.float32)).cuda()
x, wt, bias = t(1, (b, 64, h, w)), t(2, (64, 64, 3, 3), 1 / 24.0), t(3, (64,), 0.1)
gamma, beta, r, tp = 1 + t(4, (64,), 0.1), t(5, (64,), 0.1), t(6, (b, 64, h, w)), t(7, (b, 64), 0.5)
ss = ops.gn_scale_shift(x, gamma, beta, 32, 1e-5)
xb, rb, wr, wh = ops.to_blocked(x), ops.to_blocked(r), ops.relayout_conv_weight(wt), ops.relayout_conv_weight_h2(wt)
kw = dict(ksize=3, cout=64, src_blocked=True, dst_blocked=True, want_stats=True, gn_scale_shift=ss, silu=True, weight_h2=wh,
          temb=tp, temb_stride=64, residual=rb)
lib = _lib.load()
gf = 2.0 * b * h * w * 64 * 64 * 9 / 1e9
for on in (0, 1, 0, 1):
    _lib.check(lib.dsg_set_tuning(33, on))
    for _ in range(3):
        ops.conv2d_fused(xb, wr, bias, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.conv2d_fused(xb, wr, bias, **kw)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 20 * 1e6
    print(f"{os.path.basename(_lib.LIB_PATH):28s} key33={on}  {h}x{w} B={b}: {us:8.1f} us  {gf / us * 1e3:6.0f} TF/s-eq ({gf / us * 1e3 / 833:.2f})")
