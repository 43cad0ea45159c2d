"""Self-consistency of single ops of the fp32 training forward under GPU sharing (run next to background loaders: tools/race_ops.sh).
Each op runs `reps` times on the same inputs; a deterministic op gives ONE distinct checksum."""
import os, sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drivescenegen_amd import ops, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
only = sys.argv[2] if len(sys.argv) > 2 else ""
dev = "cuda"
B = 4


def t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * scale).astype(np.float32)).to(dev)


def cs(*ts):
    return " ".join(float(x.double().abs().sum()).hex()[4:14] for x in ts if x is not None)


def report(name, fn):
    if only and only not in name:
        return
    c = Counter(fn() for _ in range(reps))
    torch.cuda.synchronize()
    print(f"{name:34s} {len(c):4d} distinct of {reps}   most common {c.most_common(1)[0]}", flush=True)


def conv_case(name, c0, cout, h, w, k=3, stride=1, ups=False, gn=True, c1=0, temb=True, res=True, stats=True):
    x0 = t(1, (B, c0, h, w))
    x1 = t(2, (B, c1, h, w)) if c1 else None
    c = c0 + c1
    wt = t(3, (cout, c, k, k), 1.0 / np.sqrt(k * k * c))
    wf = ops.relayout_conv_weight(wt)
    wh = ops.relayout_conv_weight_h2(wt) if c % 16 == 0 else None
    fold = ops.relayout_conv_weight_h2_fold(wt) if ups else None
    bias = t(4, (cout,), 0.1)
    ss = None
    if gn:
        ss, _ = ops.gn_scale_shift_train(x0, 1 + t(5, (c,), 0.1), t(6, (c,), 0.1), 32 if c % 32 == 0 else 1, 1e-5, src1=x1)
    ho, wo = ((2 * h, 2 * w) if ups else (h // stride, w // stride))
    tp = t(7, (B, cout), 0.3) if temb else None
    r = t(8, (B, cout, ho, wo)) if res else None

    def run():
        y = ops.conv2d_fused(x0, wf, bias, src1=x1, ksize=k, stride=stride, upsample=ups, gn_scale_shift=ss, silu=gn,
                             temb=tp, temb_stride=cout if temb else 0, residual=r, cout=cout, weight_h2=wh,
                             weight_h2_fold=fold, want_stats=stats)
        y, s = y if stats else (y, None)
        return cs(y, s)
    report(name, run)


conv_case("conv3x3 32->32 @64 gn temb res", 32, 32, 64, 64)
conv_case("conv3x3 64->64 @32 gn temb res", 64, 64, 32, 32)
conv_case("conv3x3 cat 64+32->32 @64", 64, 32, 64, 64, c1=32)
conv_case("conv3x3 cat 64+64->64 @32", 64, 64, 32, 32, c1=64)
conv_case("conv1x1 shortcut 96->32 @64", 64, 32, 64, 64, k=1, gn=False, c1=32, temb=False, res=False, stats=False)
conv_case("conv1x1 qkv 64->192 @32", 64, 192, 32, 32, k=1, gn=True, temb=False, res=False, stats=False)
conv_case("conv stride2 32->32 @64", 32, 32, 64, 64, stride=2, gn=False, temb=False, res=False, stats=False)
conv_case("conv ups 64->64 @32", 64, 64, 32, 32, ups=True, gn=False, temb=False, res=False)
conv_case("conv_in 3->32 @64", 3, 32, 64, 64, gn=False, temb=False, res=False)
conv_case("conv_out 32->3 @64", 32, 3, 64, 64, temb=False, res=False, stats=False)
x = t(11, (B, 64, 32, 32))
g, b = 1 + t(12, (64,), 0.1), t(13, (64,), 0.1)
report("gn_scale_shift_train 64 @32", lambda: cs(*ops.gn_scale_shift_train(x, g, b, 32, 1e-5)))
qkv = t(14, (B, 192, 1024))
report("attention_train 8 heads L=1024", lambda: cs(*ops.attention_train(qkv, 8)))
a, wl, bl = t(15, (B, 128)), t(16, (512, 128), 0.1), t(17, (512,), 0.1)
report("linear 128->512", lambda: cs(ops.linear(a, wl, bl)))
