"""A background neighbour for tools/probes/run_pk_opsel.sh: loops ONE op of the library forever.  Usage: neighbour.py <op>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import drivescenegen_amd as d
from drivescenegen_amd import ops, synth
op = sys.argv[1]
dev = "cuda"
def t(seed, shape, scale=1.0):
    return torch.from_numpy((synth.normal(seed, shape) * scale).astype(np.float32)).to(dev)
B = 16
if op in ("conv3x3", "conv3x3_bf16", "conv1x1"):
    mode = "bf16" if op.endswith("bf16") else "fp32"
    k = 1 if op == "conv1x1" else 3
    x = ops.to_blocked(t(1, (B, 256, 64, 64)), mode)
    wt = t(2, (256, 256, k, k), 0.02)
    wh = ops.relayout_conv_weight_h2(wt) if mode == "fp32" else ops.pack_conv_weight(wt, ops.PACK_FWD, mode)
    ss = torch.stack([1 + t(3, (B, 256), 0.1), t(4, (B, 256), 0.1)], -1).contiguous()
    f = lambda: ops.conv2d_fused(x, None, None, ksize=k, cout=256, gn_scale_shift=ss if k == 3 else None, silu=k == 3, src_blocked=True,
                                 dst_blocked=True, weight_h2=wh, compute_dtype=mode)
elif op == "conv_f32mfma":   # the exact f32 MFMA kernel ([N,C,H,W], no split weights)
    x = t(1, (B, 64, 64, 64)); wf = ops.relayout_conv_weight(t(2, (64, 64, 3, 3), 0.04))
    f = lambda: ops.conv2d_fused(x, wf, None, ksize=3, cout=64)
elif op == "attention":
    qkv = ops.to_blocked(t(1, (B, 3 * 512, 32, 32)))
    f = lambda: ops.attention_blocked(qkv, 64) if hasattr(ops, "attention_blocked") else ops.attention(t(1, (B, 1536, 1024)), 64)
elif op == "gn_stats":
    x = ops.to_blocked(t(1, (B, 128, 128, 128)))
    f = lambda: ops.gn_channel_stats_blocked(x, splits=8)
elif op == "layout":
    x = t(1, (B, 128, 128, 128))
    f = lambda: ops.to_blocked(x)
elif op == "ddim":
    sch = d.DDIMScheduler(); sch.set_timesteps(50)
    x = t(1, (B, 4, 256, 256)); e = t(2, (B, 4, 256, 256))
    f = lambda: sch.step(e, 500, x).prev_sample
else:
    raise SystemExit("unknown op " + op)
while True:
    for _ in range(50):
        f()
    torch.cuda.synchronize()
