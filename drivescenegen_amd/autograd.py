"""Differentiable (training) forward of ``UNet2DModel`` on the HIP engine.

Reference: /root/reference/DriveSceneGen/pipeline/training_pipeline.py:84-86
    noise_pred = model(noisy_pattern, timesteps, return_dict=False)[0]
    loss = F.mse_loss(noise_pred, noise);  accelerator.backward(loss)

torch.autograd sees ONE node for the whole network (``_UNetTrainFn``): its forward runs the fused per-op
C-ABI calls of ops.py and records a tape; its backward walks the tape in reverse with our backward
kernels and ACCUMULATES parameter gradients straight into a flat fp32 slab whose slices are the
parameters' ``.grad`` tensors (one buffer -> one global-norm kernel, one fused AdamW kernel, bucketed
all-reduce on plain slices).  Nothing but conv outputs, GroupNorm statistics and the attention
log-sum-exp is kept for backward: GroupNorm-apply + SiLU are recomputed inside the weight-gradient
gather, exactly as they are folded into the forward gather.
"""
from __future__ import annotations

import math
import os

import torch

from . import _lib, ops

_EAGER_RELAYOUT = os.environ.get("DSG_EAGER_RELAYOUT") == "1"  # A/B switch of TrainState.lazy_w (tools/train_bench.py)


SLABS = {}  # storage data_ptr -> (flat gradient slab, number of parameter slices): lets the optimizer / clipper
            # recognise slab-backed gradients and tell the whole parameter set from a subset


class TrainState:
    """Flat gradient slab + engine-layout weight copies of one UNet2DModel (created on first training use)."""

    def __init__(self, model):
        self.model = model
        self.items = list(model.state_dict(keep_vars=True).items())
        dev = model.device
        sizes = [p.numel() for _, p in self.items]
        self.offsets = {}
        off = 0
        for (name, _), n in zip(self.items, sizes):
            self.offsets[name] = (off, n)
            off += (n + 63) // 64 * 64  # 256-B aligned slices
        self.total = off
        self.grad_flat = torch.zeros(off, dtype=torch.float32, device=dev)
        SLABS[self.grad_flat.untyped_storage().data_ptr()] = (self.grad_flat, len(self.items))
        self.params = dict(self.items)
        self.wf, self.wd, self.versions = {}, {}, {}
        self.wh, self.whd = {}, {}  # fp16x2-split copies (forward / data-gradient) of the eligible 3x3 weights
        self.scratch16 = {}         # padded 16-bit buffers of the mixed-precision tape's conv_in / conv_out weight gradients
        self.needs_w = {}           # (name, "wf" | "wd") -> does the mixed-precision tape's call read the fp32 engine layout?
        self.freqs = ops.sinusoid_freqs(model.config.block_out_channels[0]).to(dev)
        self.grad_ready_hooks = []  # callables(name) fired when a parameter's gradient is final (DDP buckets)
        self.post_backward = []     # callables() run when the backward walk is complete, before the internal loss scale
        #                             is taken back out of the slab (the bucketed all-reduce finishes here)
        self.packs16 = {}           # dsg_dtype -> _Packs: 16-bit operand images for the mixed-precision tape
        self.fuse_cache = {}        # (resnet, shapes, dtype, tuning epoch) -> does conv2 take the 1x1 shortcut into its K loop?
        self.attach()

    def grad(self, name):
        off, n = self.offsets[name]
        return self.grad_flat[off:off + n]

    def attach(self):
        """(Re)point every parameter's .grad at its slice of the slab; a slice whose .grad was dropped
        (optimizer.zero_grad(set_to_none=True)) is zeroed first."""
        for name, p in self.items:
            g = self.grad(name).view(p.shape)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                if p.grad is not None:
                    g.copy_(p.grad)
                else:
                    g.zero_()
                p.grad = g

    # engine-layout weights, refreshed when the parameter changed (every optimizer step)
    def _fresh(self, key):
        """key: a parameter name, or (name, variant) when several cached forms of one parameter are tracked apart"""
        p = self.params[key[0] if isinstance(key, tuple) else key]
        sig = (p.data_ptr(), p._version)
        if self.versions.get(key) == sig:
            return False
        self.versions[key] = sig
        return True

    def conv_w(self, name, split=True):
        """(forward, data-gradient) fp32 engine layouts of a conv weight; split: also its fp16x2-split packs (the fp32
        tape's matrix-core operands -- the mixed-precision tape keeps its own 16-bit packs and skips them)."""
        if self._fresh((name, split)) or name not in self.wf:
            p = self.params[name].detach()
            self.wf[name] = ops.relayout_conv_weight(p, out=self.wf.get(name))
            cin = p.shape[1]
            if name not in self.wd:
                k = p.shape[2] if p.dim() == 4 else 1
                self.wd[name] = torch.zeros((p.shape[0], k * k, ops._pad32(cin) + 64), dtype=torch.float32,
                                            device=p.device)
            ops.relayout_conv_weight_dgrad(p, out=self.wd[name])
            cout = p.shape[0]
            if not split:
                return self.wf[name], self.wd[name]
            if cin % 16 == 0 and cout % 64 == 0:   # forward conv on the fp16x2-split matrix-core path
                self.wh[name] = ops.relayout_conv_weight_h2(p, out=self.wh.get(name))
            if cout % 16 == 0 and cin % 64 == 0:   # its data gradient (K = cout, N = cin)
                self.whd[name] = ops.relayout_conv_weight_h2_dgrad(p, out=self.whd.get(name))
        return self.wf[name], self.wd[name]

    def wf_of(self, name):
        """fp32 engine layout of a conv weight for the forward kernels that read it (conv_in, conv_out, narrow maps)."""
        if self._fresh((name, "wf")) or name not in self.wf:
            self.wf[name] = ops.relayout_conv_weight(self.params[name].detach(), out=self.wf.get(name))
        return self.wf[name]

    def wd_of(self, name):
        """... and its data-gradient form."""
        if self._fresh((name, "wd")) or name not in self.wd:
            p = self.params[name].detach()
            if name not in self.wd:
                k = p.shape[2] if p.dim() == 4 else 1
                self.wd[name] = torch.zeros((p.shape[0], k * k, ops._pad32(p.shape[1]) + 64), dtype=torch.float32, device=p.device)
            ops.relayout_conv_weight_dgrad(p, out=self.wd[name])
        return self.wd[name]

    def h2_of(self, name):
        """(forward, data-gradient) fp16x2-split operand images of a conv weight for the fp32 tape, or None where the split
        kernels do not take the shape; refreshed when the parameter changes"""
        if self._fresh((name, "h2")) or (name not in self.wh and name not in self.whd):
            p = self.params[name].detach()
            cout, cin = p.shape[0], p.shape[1]
            if cin % 16 == 0 and cout % 64 == 0:
                self.wh[name] = ops.relayout_conv_weight_h2(p, out=self.wh.get(name))
            if cout % 16 == 0 and cin % 64 == 0:
                self.whd[name] = ops.relayout_conv_weight_h2_dgrad(p, out=self.whd.get(name))
        return self.wh.get(name), self.whd.get(name)

    def pack32(self, name, kind):
        """fp16x2-split operand image of `kind` (ops.PACK_*) of a conv weight for the fp32 tape, refreshed when it changes"""
        key = (name, "p32", kind)
        if self._fresh(key) or key not in self.wh:
            self.wh[key] = ops.pack_conv_weight(self.params[name].detach(), kind, 0, out=self.wh.get(key))
        return self.wh[key]

    def lazy_w(self, name, which, call):
        """call(w): w = the fp32 engine layout (`which`: "wf" | "wd") only if the library's kernels for this call read it.
        Most convs of the mixed-precision tape run on their 16-bit operand images alone; re-laying out all 70 weights twice
        per step was 1.5 ms of launches.  The first call passes None; the library refuses it BEFORE launching anything when
        the call does need the layout, and the answer is remembered."""
        getter = self.wf_of if which == "wf" else self.wd_of
        need = self.needs_w.get((name, which))
        if need or _EAGER_RELAYOUT:
            return call(getter(name))
        try:
            out = call(None)
        except _lib.DsgError as e:
            if "weight is NULL" not in str(e):
                raise
            self.needs_w[(name, which)] = True
            return call(getter(name))
        self.needs_w[(name, which)] = False
        return out

    def qkv_w(self, prefix):
        """Fused q/k/v projection: forward [C][1][3C], data-gradient [3C][1][C(+pad)], bias [3C]."""
        names = [f"{prefix}.{t}.weight" for t in ("to_q", "to_k", "to_v")]
        key = prefix + ".qkv"
        fresh = any([self._fresh(n) for n in names])
        if fresh or key not in self.wf:
            ws = [self.params[n].detach() for n in names]
            c = ws[0].shape[0]
            if key not in self.wf:
                dev = ws[0].device
                self.wf[key] = torch.zeros((c, 1, 3 * c), dtype=torch.float32, device=dev)
                self.wd[key] = torch.zeros((3 * c, 1, ops._pad32(c) + 64), dtype=torch.float32, device=dev)
                self.wf[key + ".bias"] = torch.zeros(3 * c, dtype=torch.float32, device=dev)
            h2 = c % 64 == 0 and not _R5_ROUTES   # the fp16x2-split operand images of the fused projection and of its data gradient (the fp32 tape:
            if h2 and key not in self.wh:   # round 6 -- the two calls ran on the exact f32 kernel at 77 TF/s, 1.34 ms each at B=64)
                self.wh[key] = torch.zeros((c // 16, 2, 1, 2, 3 * c, 8), dtype=torch.float16, device=ws[0].device)
                self.whd[key] = torch.zeros((3 * c // 16, 2, 1, 2, c, 8), dtype=torch.float16, device=ws[0].device)
            for i, w in enumerate(ws):
                ops.relayout_conv_weight(w, out=self.wf[key], cout_total=3 * c, cout_off=i * c)
                ops.relayout_conv_weight_dgrad(w, out=self.wd[key][i * c:(i + 1) * c])
                if h2:
                    ops.relayout_conv_weight_h2(w, out=self.wh[key], cout_total=3 * c, cout_off=i * c)
                    # (the data gradient's K is the 3C channels of dqkv: the three projections' images, stacked along K)
                    ops.relayout_conv_weight_h2_dgrad(w, out=self.whd[key][i * c // 16:(i + 1) * c // 16])
        # biases are tiny: refresh unconditionally through our own copy kernel-free path (slice assignment is a
        # device memcpy, not arithmetic)
        c = self.params[names[0]].shape[0]
        for i, t in enumerate(("to_q", "to_k", "to_v")):
            self.wf[key + ".bias"][i * c:(i + 1) * c].copy_(self.params[f"{prefix}.{t}.bias"].detach())
        return self.wf[key], self.wd[key], self.wf[key + ".bias"]


def get_train_state(model) -> TrainState:
    st = getattr(model, "_train_state", None)
    if st is None or st.grad_flat.device != model.device:
        st = TrainState(model)
        model._train_state = st
    return st


# A/B hook of the measurement scripts (tools/collect_r06.sh): DSG_F32_TAPE_R5=1 keeps the fp32 tape on the routes round 5 ran -- the
# fused q/k/v projection on the exact f32 kernel, conv_in / conv_out weight gradients unpadded, residual + skip gradients summed by a
# pass of their own -- next to DSG_TUNING="37=0,38=0" for the two kernel-level changes.
_R5_ROUTES = os.environ.get("DSG_F32_TAPE_R5") == "1"
# ... and DSG_UPS_DGRAD_FULLRES=1 keeps both tapes' up-sampler data gradient on the full-resolution 3x3 conv + 2x2 sums
_UPS_DGRAD_FULLRES = os.environ.get("DSG_UPS_DGRAD_FULLRES") == "1"


class _Tape:
    def __init__(self):
        self.recs = []
        self.grads = {}
        self.pend = {}   # id(tensor) -> (gradient, second contribution) not summed yet (addg(lazy=True))

    def g(self, t):
        pair = self.pend.pop(id(t), None) if self.pend else None
        if pair is not None:
            self.grads[id(t)] = ops.add(pair[0], pair[1])
        return self.grads.get(id(t))

    def setg(self, t, g):
        self.grads[id(t)] = g

    def addg(self, t, g, lazy=False):
        """Fan-in: an existing gradient and a new contribution are summed out of place.  lazy (the 16-bit tape's residual
        branch): the pair is kept as it is -- the GroupNorm backward that runs next on `t` adds both in its own pass (g2) --
        and summed only if somebody else asks for the gradient first (g)."""
        cur = self.g(t)
        if cur is not None and lazy:
            self.pend[id(t)] = (cur, g)
            self.grads.pop(id(t), None)
            return
        self.grads[id(t)] = g if cur is None else ops.add(cur, g)

    def g2(self, t):
        """(gradient, second term or None) of `t` without materialising a pending pair"""
        pair = self.pend.pop(id(t), None)
        return pair if pair is not None else (self.grads.get(id(t)), None)


def _forward(model, st: TrainState, tape: _Tape, sample, timesteps):
    cfg = model.config
    P = st.params
    groups, eps = cfg.norm_num_groups, cfg.norm_eps
    proj_total = 0
    resnets = []  # (prefix, cin, cout, toff) in forward order, to lay out the time_emb_proj matrix

    def walk():
        for i, blk in enumerate(model.down_blocks):
            for j in range(len(blk.resnets)):
                yield f"down_blocks.{i}.resnets.{j}"
        yield "mid_block.resnets.0"
        yield "mid_block.resnets.1"
        for i, blk in enumerate(model.up_blocks):
            for j in range(len(blk.resnets)):
                yield f"up_blocks.{i}.resnets.{j}"

    toffs = {}
    for pre in walk():
        cout = P[pre + ".time_emb_proj.bias"].numel()
        toffs[pre] = proj_total
        proj_total += cout
        resnets.append(pre)

    # ---- timestep path ----
    w1, b1 = P["time_embedding.linear_1.weight"].detach(), P["time_embedding.linear_1.bias"].detach()
    w2, b2 = P["time_embedding.linear_2.weight"].detach(), P["time_embedding.linear_2.bias"].detach()
    act, emb, z1, z2 = ops.time_embed_train(timesteps, w1, b1, w2, b2, st.freqs)
    dim = w1.shape[0]
    key = "_tproj"
    if key not in st.wf:
        st.wf[key] = torch.empty((proj_total, dim), dtype=torch.float32, device=sample.device)
        st.wf[key + ".bias"] = torch.empty(proj_total, dtype=torch.float32, device=sample.device)
    wp, bp = st.wf[key], st.wf[key + ".bias"]
    dsts, srcs = [], []   # gather the 22 matrices into one: device copies only, in ONE multi-tensor launch (44 memcpys before)
    for pre in resnets:
        o = toffs[pre]
        w = P[pre + ".time_emb_proj.weight"].detach()
        dsts += [wp[o:o + w.shape[0]], bp[o:o + w.shape[0]]]
        srcs += [w, P[pre + ".time_emb_proj.bias"].detach()]
    torch._foreach_copy_(dsts, srcs)
    tproj = ops.linear(act, wp, bp)
    dtproj = torch.zeros_like(tproj)
    tape.temb = dict(act=act, emb=emb, z1=z1, z2=z2, wp=wp, dtproj=dtproj, toffs=toffs, resnets=resnets, w2=w2)

    # ---- fused conv op with tape record ----
    pstats = {}  # id(tensor) -> per-tile GroupNorm statistics written by the conv that produced it

    def norm_ss(x0, x1, gn):
        """(scale_shift, mean_rstd) of GroupNorm `gn` over cat(x0, x1): from the producers' epilogue statistics where
        every source has them, else by a statistics pass."""
        g, b = P[gn + ".weight"].detach(), P[gn + ".bias"].detach()
        s0 = pstats.get(id(x0))
        s1 = pstats.get(id(x1)) if x1 is not None else None
        if s0 is not None and (x1 is None or s1 is not None):
            return ops.gn_scale_shift_from_parts_train(s0, g, b, groups, eps, x0.shape[2] * x0.shape[3], stats1=s1)
        return ops.gn_scale_shift_train(x0, g, b, groups, eps, src1=x1)

    bounds = {}

    def bound_of(x):
        st_x = pstats.get(id(x))
        if st_x is None:
            return None
        if id(x) not in bounds:
            bounds[id(x)] = ops.range_bound_from_stats(st_x)
        return bounds[id(x)]

    def conv(x0, wname, x1=None, gn=None, silu=False, k=3, stride=1, ups=False, toff=None, res=None,
             need_dx=True, feeds_norm=False):
        wh, whd = st.h2_of(wname + ".weight")
        bias = P[wname + ".bias"].detach()
        cout = bias.numel()
        ss = mr = None
        if gn is not None:
            ss, mr = norm_ss(x0, x1, gn)
        # (up-sampler convs: the folded 2x2 phase kernels of the inference plan instead of the nearest-x2 gather)
        fold = st.pack32(wname + ".weight", ops.PACK_FOLD) if (ups and k == 3 and x1 is None and gn is None
                                                                 and x0.shape[1] % 16 == 0 and cout % 64 == 0) else None
        # (down-sampler convs: the 2x2 conv over the space-to-depth image, as the inference plan's -- [N,C,H,W] tensors here)
        s2p = st.pack32(wname + ".weight", ops.PACK_S2) if (stride == 2 and k == 3 and x1 is None and gn is None and not _R5_ROUTES
                                                             and x0.shape[1] % 8 == 0 and cout % 8 == 0 and x0.shape[2] % 16 == 0
                                                             and (x0.shape[3] % 64 == 0 or x0.shape[3] in (16, 32))) else None
        # Range guard of the split path: a conv WITHOUT a norm in front (shortcut, up- / down-sampler) reads the
        # residual stream as it is; where the producing conv left statistics, their per-image bound goes along
        # (dsg_conv_args.src_bound: exact power-of-two pre-scaling outside [2^-6, 2^12], a no-op inside)
        b0 = b1 = None
        if gn is None and (wh is not None or fold is not None or s2p is not None):
            b0 = bound_of(x0)
            b1 = bound_of(x1) if (x1 is not None and b0 is not None) else None
            if x1 is not None and b1 is None:
                b0 = None
        # (wf, the fp32 engine layout, only for the calls whose kernels read it: TrainState.lazy_w)
        y = st.lazy_w(wname + ".weight", "wf", lambda wf: ops.conv2d_fused(
            x0, wf, bias, src1=x1, ksize=k, stride=stride, upsample=ups, gn_scale_shift=ss, silu=silu,
            temb=None if toff is None else tproj[:, toff:], temb_stride=tproj.stride(0), residual=res, cout=cout,
            weight_h2=wh, weight_h2_fold=fold, weight_h2_s2=s2p, want_stats=feeds_norm, src_bound=b0, src_bound1=b1))
        if feeds_norm:
            y, ystats = y
            if ystats is not None:
                pstats[id(y)] = ystats
        tape.recs.append(dict(kind="conv", x0=x0, x1=x1, ss=ss, mr=mr, gn=gn, silu=silu, k=k, stride=stride, ups=ups,
                              toff=toff, res=res, y=y, wname=wname, cout=cout, need_dx=need_dx, whd=whd))
        return y

    def resnet(x, skip, pre):
        h = conv(x, pre + ".conv1", x1=skip, gn=pre + ".norm1", silu=True, toff=toffs[pre], feeds_norm=True)
        if (pre + ".conv_shortcut.weight") in P:
            sc = conv(x, pre + ".conv_shortcut", x1=skip, k=1)
        else:
            sc = x
        return conv(h, pre + ".conv2", gn=pre + ".norm2", silu=True, res=sc, feeds_norm=True)

    def attention(x, pre):
        wf, wd, bias = st.qkv_w(pre)
        c = x.shape[1]
        heads = c // cfg.attention_head_dim
        gnn = pre + ".group_norm"
        ss, mr = norm_ss(x, None, gnn)
        qkv = ops.conv2d_fused(x, wf, bias, ksize=1, gn_scale_shift=ss, silu=False, weight_h2=st.wh.get(pre + ".qkv"))
        n, _, hh, ww = x.shape
        o, lse = ops.attention_train(qkv.view(n, 3 * c, hh * ww), heads)
        o = o.view(n, c, hh, ww)
        tape.recs.append(dict(kind="qkv", x=x, ss=ss, mr=mr, gn=gnn, pre=pre, qkv=qkv, wd=wd, whd=st.whd.get(pre + ".qkv")))
        tape.recs.append(dict(kind="attn", qkv=qkv, o=o, lse=lse, heads=heads))
        return conv(o, pre + ".to_out.0", k=1, res=x, feeds_norm=True)

    x = conv(sample, "conv_in", need_dx=False)
    skips = [x]
    for i, blk in enumerate(model.down_blocks):
        pre = f"down_blocks.{i}"
        for j in range(len(blk.resnets)):
            x = resnet(x, None, f"{pre}.resnets.{j}")
            if hasattr(blk, "attentions"):
                x = attention(x, f"{pre}.attentions.{j}")
            skips.append(x)
        if hasattr(blk, "downsamplers"):
            x = conv(x, f"{pre}.downsamplers.0.conv", stride=2)
            skips.append(x)
    x = resnet(x, None, "mid_block.resnets.0")
    if hasattr(model.mid_block, "attentions"):
        x = attention(x, "mid_block.attentions.0")
    x = resnet(x, None, "mid_block.resnets.1")
    for i, blk in enumerate(model.up_blocks):
        pre = f"up_blocks.{i}"
        for j in range(len(blk.resnets)):
            s = skips.pop()
            x = resnet(x, s, f"{pre}.resnets.{j}")
            if hasattr(blk, "attentions"):
                x = attention(x, f"{pre}.attentions.{j}")
        if hasattr(blk, "upsamplers"):
            x = conv(x, f"{pre}.upsamplers.0.conv", ups=True, feeds_norm=True)
    return conv(x, "conv_out", gn="conv_norm_out", silu=True)


def _wgrad_padded_f32(st, x0, x1, dy, wname, k, rec):
    """Weight gradient of a plain 3x3 conv with FEW channels on one side (conv_in: 3 / 4 / 8 inputs; conv_out: as many outputs) on
    the fp32 tape: the narrow side is padded with zero channels to the split kernel's granule (32 inputs, 64 outputs) -- a
    [64 x 32..64] weight gradient of which one corner is kept.  The exact f32 kernels took 2.9 ms (conv_out) + 1.4 ms (conv_in) of
    the configs[2] step at 50-60 TF/s; the padded buffers keep their zero channels between steps (only the real channels are
    copied in).  Returns False for shapes this does not serve (the caller then runs the plain call)."""
    if k != 3 or rec["stride"] != 1 or rec["ups"] or x1 is not None or _R5_ROUTES:
        return False
    n, cin, h, w = x0.shape
    cout = dy.shape[1]
    cp, op = (cin + 31) // 32 * 32, (cout + 63) // 64 * 64
    if (cp == cin and op == cout) or cp > 64 or op > 64 or not ops.wgrad_h2_supported(cp, 0, op, h, w, 3, 1, False):
        return False

    def padded(t, c, cpad, tag):
        if c == cpad:
            return t
        key = ("pad32", wname, tag, n, cpad, h, w)
        buf = st.scratch16.get(key)
        if buf is None:
            buf = st.scratch16[key] = torch.zeros((n, cpad, h, w), dtype=torch.float32, device=t.device)
        buf[:, :c].copy_(t)     # (a device copy of the real channels; the zero channels were written once)
        return buf
    xb, dyb = padded(x0, cin, cp, "x"), padded(dy.contiguous(), cout, op, "dy")
    ss = rec["ss"]
    if ss is not None and cp != cin:
        ssp = torch.zeros((n, cp, 2), dtype=ss.dtype, device=ss.device)   # (zero scale / shift: a zero channel stays zero)
        ssp[:, :cin].copy_(ss)
        ss = ssp
    key = ("pad32", wname, "dw")
    tmp = st.scratch16.get(key)
    if tmp is None:
        tmp = st.scratch16[key] = torch.empty((op, cp, 3, 3), dtype=torch.float32, device=dy.device)
    tmp.zero_()
    ops.conv_wgrad(xb, dyb, tmp, ksize=3, gn_scale_shift=ss, silu=rec["silu"], cout=op)
    g = st.grad(wname + ".weight")
    g.copy_(ops.add(g, tmp[:cout, :cin].contiguous()))
    return True


def _backward(model, st: TrainState, tape: _Tape, dout):
    P = st.params
    groups = model.config.norm_num_groups
    tb = tape.temb
    fire = st.grad_ready_hooks

    def done(*names):
        for h in fire:
            for nm in names:
                h(nm)

    dysums = {}   # id(dy) -> (dy, its per-(n, c) sums, row stride): filled by every conv's backward, read by the shortcut's
    for rec in reversed(tape.recs):
        kind = rec["kind"]
        if kind == "conv":
            dy = tape.g(rec["y"])
            if dy is None:
                continue
            wname, cout, k = rec["wname"], rec["cout"], rec["k"]
            x0, x1 = rec["x0"], rec["x1"]
            if rec["res"] is not None:
                tape.addg(rec["res"], dy, lazy=not _R5_ROUTES)   # (a pending pair is summed inside the next GroupNorm backward on it: add0b)
            # up-sampler conv (no norm, one source): both gradients run at full resolution on the split matrix-core
            # kernels -- the weight gradient from the materialised nearest-x2 input, the data gradient as a plain
            # transposed conv followed by the upsample's adjoint (2x2 sum-pool)
            ups_h2 = bool(rec["ups"]) and rec["whd"] is not None and x1 is None and rec["gn"] is None and k == 3
            # bias (+ time-embedding) gradient: per-(n, c) sums of dy -- a by-product of the split weight-gradient kernel
            # where it serves the conv (the dY tiles pass through it anyway), a pass of their own over dy otherwise
            if rec["toff"] is not None:
                sums, sstride = tb["dtproj"][:, rec["toff"]:], tb["dtproj"].stride(0)
            else:
                sums = torch.empty((dy.shape[0], cout), dtype=torch.float32, device=dy.device)
                sstride = cout
            byp = ops.wgrad_h2_supported(x0.shape[1], 0 if (x1 is None or ups_h2) else x1.shape[1], cout, dy.shape[2], dy.shape[3],
                                         k, 1 if ups_h2 else rec["stride"], False if ups_h2 else bool(rec["ups"]))
            # (with the sums, the same pass adds their total over the batch to the bias gradient: dy_bias_grad)
            kw = dict(dy_sums=sums, dy_sums_stride=sstride, bias_grad=st.grad(wname + ".bias")) if byp else {}
            if ups_h2:
                ops.conv_wgrad(ops.upsample_nearest2x(x0), dy, st.grad(wname + ".weight"), ksize=k, **kw)
            elif not byp and _wgrad_padded_f32(st, x0, x1, dy, wname, k, rec):
                pass   # conv_in / conv_out: few channels on one side, padded with zero channels onto the split kernel
            else:
                ops.conv_wgrad(x0, dy, st.grad(wname + ".weight"), src1=x1, ksize=k, stride=rec["stride"],
                               upsample=rec["ups"], gn_scale_shift=rec["ss"], silu=rec["silu"], **kw)
            if not byp:
                # (a resnet's conv_shortcut sees the very gradient tensor its conv2 saw -- y = conv2(..) + shortcut(x) -- whose
                #  per-(n, c) sums conv2's weight gradient has just left behind: no pass of its own over dy for the bias gradient)
                seen = None if _R5_ROUTES else dysums.get(id(dy))
                if seen is not None and seen[0] is dy:
                    sums, sstride = seen[1], seen[2]
                else:
                    ops.channel_sums(dy, out=sums, out_stride=sstride)
                ops.reduce_rows_add(sums, st.grad(wname + ".bias"), stride=sstride)
            dysums[id(dy)] = (dy, sums, sstride)
            done(wname + ".weight", wname + ".bias")
            if rec["toff"] is not None:
                _temb_proj_grads(st, tb, wname[:-len(".conv1")], done)
            if not rec["need_dx"]:
                continue
            cin0, cin1 = x0.shape[1], (x1.shape[1] if x1 is not None else 0)
            wdn, wd_stride = wname + ".weight", ops._pad32(cin0 + cin1) + 64  # (row length of TrainState.wd_of's layout)
            # up-sampler conv: one 4x4 stride-2 window over dY per low-resolution pixel (16 taps: dsg_conv_args.s2_window4) where the
            # space-to-depth kernel takes the shape; else the 3x3 data gradient at full resolution + 2x2 sums (36 taps)
            if (rec["ups"] and x1 is None and rec["gn"] is None and k == 3 and not _UPS_DGRAD_FULLRES and not _R5_ROUTES
                    and cout % 8 == 0 and cin0 % 8 == 0 and x0.shape[2] % 8 == 0 and (x0.shape[3] % 32 == 0 or x0.shape[3] in (8, 16))):
                tape.setg(x0, ops.conv2d_fused(dy, None, ksize=3, stride=2, cout=cin0, residual=tape.g(x0), s2_window4=True,
                                               weight_h2_s2=st.pack32(wdn, ops.PACK_DGRAD_UPS)))
                continue
            if ups_h2:
                dfull = st.lazy_w(wdn, "wd", lambda wd: ops.conv2d_fused(dy, wd, ksize=k, cout=cin0, weight_h2=rec["whd"]))
                tape.setg(x0, ops.sumpool2x2(dfull, add=tape.g(x0)))
                continue
            up_mode = 2 if rec["stride"] == 2 else 0
            whd = rec["whd"] if (rec["stride"] == 1 and not rec["ups"]) else None  # the split kernel has no pool / zero-stuff mode
            if rec["gn"] is not None:
                # the norm's backward statistics (sum du, sum du * x per tile) come out of the data-gradient conv's epilogue where
                # its kernel has that form (dsg_conv_args.gnb_*): the pass over x and dA that computed them is not run
                gnb = dict(x0=x0, x1=x1, ss=rec["ss"], silu=rec["silu"]) if (whd is not None and up_mode == 0 and not rec["ups"]) else None

                def dgrad_gn(wd):
                    kw = dict(ksize=k, upsample=up_mode, cout=cin0 + cin1, pool2=rec["ups"], weight_h2=whd)
                    if gnb is not None and ops.conv2d_fused(dy, wd, gnb=dict(gnb, query_only=True), **kw):
                        return ops.conv2d_fused(dy, wd, gnb=gnb, want_stats=True, **kw)
                    return ops.conv2d_fused(dy, wd, **kw), None
                da, parts = st.lazy_w(wdn, "wd", dgrad_gn)
                gnn = rec["gn"]
                a0, a0b = tape.g2(x0)
                dx0, dx1 = ops.gn_bwd(x0, da, rec["ss"], rec["mr"], P[gnn + ".weight"].detach(), groups, rec["silu"],
                                      st.grad(gnn + ".weight"), st.grad(gnn + ".bias"), src1=x1, add0=a0, add0b=a0b,
                                      add1=tape.g(x1) if x1 is not None else None, parts=parts)
                done(gnn + ".weight", gnn + ".bias")
                tape.setg(x0, dx0)
                if x1 is not None:
                    tape.setg(x1, dx1)
            else:
                # no norm in front: the data gradient lands on the source(s) directly (one conv per source,
                # selecting that source's columns of the transposed weight; the fan-in add rides the epilogue)
                s2fold = (rec["stride"] == 2 and k == 3 and x1 is None and dy.shape[1] % 16 == 0 and cin0 % 64 == 0
                          and dy.shape[3] % 32 == 0 and dy.shape[2] % 8 == 0)
                if s2fold:  # the adjoint of the stride-2 conv = four 2x2 phase convs of dY: the folded up-sampler's kernel
                    tape.setg(x0, st.lazy_w(wdn, "wd", lambda wd: ops.conv2d_fused(
                        dy, wd, ksize=3, upsample=True, cout=cin0, residual=tape.g(x0),
                        weight_h2_fold=st.pack32(wname + ".weight", ops.PACK_DGRAD_S2))))
                else:
                    tape.setg(x0, st.lazy_w(wdn, "wd", lambda wd: ops.conv2d_fused(
                        dy, wd, ksize=k, upsample=up_mode, cout=cin0, pool2=rec["ups"], residual=tape.g(x0), weight_h2=whd,
                        wstride=wd_stride if x1 is not None else None)))
                if x1 is not None:
                    tape.setg(x1, st.lazy_w(wdn, "wd", lambda wd: ops.conv2d_fused(
                        dy, None if wd is None else wd[:, :, cin0:], ksize=k, upsample=up_mode, cout=cin1, pool2=rec["ups"],
                        residual=tape.g(x1), wstride=wd_stride,
                        weight_h2=whd if (cin0 % 8 == 0 and cin1 % 64 == 0) else None,
                        weight_h2_col=cin0)))  # (a window must end inside its row)
        elif kind == "attn":
            do = tape.g(rec["o"])
            qkv = rec["qkv"]
            n, c3, hh, ww = qkv.shape
            dqkv = ops.attention_bwd(qkv.view(n, c3, hh * ww), rec["o"].view(n, c3 // 3, hh * ww),
                                     do.view(n, c3 // 3, hh * ww), rec["lse"], rec["heads"])
            tape.setg(qkv, dqkv.view(n, c3, hh, ww))
        elif kind == "qkv":
            dqkv = tape.g(rec["qkv"])
            x, pre = rec["x"], rec["pre"]
            c = x.shape[1]
            sums = ops.channel_sums(dqkv)
            for i, t in enumerate(("to_q", "to_k", "to_v")):
                ops.reduce_rows_add(sums[:, i * c:], st.grad(f"{pre}.{t}.bias"), stride=sums.stride(0))
                ops.conv_wgrad(x, dqkv, st.grad(f"{pre}.{t}.weight"), ksize=1, gn_scale_shift=rec["ss"], silu=False,
                               cout=c, dy_coff=i * c)
                done(f"{pre}.{t}.weight", f"{pre}.{t}.bias")
            da = ops.conv2d_fused(dqkv, rec["wd"], ksize=1, cout=c, weight_h2=rec.get("whd"))
            gnn = rec["gn"]
            dx, _ = ops.gn_bwd(x, da, rec["ss"], rec["mr"], P[gnn + ".weight"].detach(), groups, False,
                               st.grad(gnn + ".weight"), st.grad(gnn + ".bias"), add0=tape.g(x))
            done(gnn + ".weight", gnn + ".bias")
            tape.setg(x, dx)

    _backward_temb(st, tb, done)


def _temb_proj_grads(st, tb, pre, done):
    """time_emb_proj gradients of ONE resnet, taken the moment its conv1's per-(n, cout) sums of dY -- this resnet's columns
    of d(time projection) -- are final.  Inside the walk on purpose: every resnet owns a time_emb_proj, so with these
    gradients left to the end of backward no gradient bucket of the data-parallel all-reduce (training.GradBuckets) was
    complete before the walk had finished, and nothing overlapped (found by tests/test_gpu_rccl_one_rank.py on the
    56.6 M-parameter network: launch order 0, 1, 2, ... from finish() instead of N-1, N-2, ... from inside the walk)."""
    if pre in tb.setdefault("proj_done", set()):
        return
    tb["proj_done"].add(pre)
    dtproj, o = tb["dtproj"], tb["toffs"][pre]
    w = st.params[pre + ".time_emb_proj.weight"]
    ops.linear_bwd(tb["act"], w.detach(), dtproj[:, o:], st.grad(pre + ".time_emb_proj.weight"),
                   st.grad(pre + ".time_emb_proj.bias"), need_dx=False, dy_stride=dtproj.stride(0))
    done(pre + ".time_emb_proj.weight", pre + ".time_emb_proj.bias")


def _backward_temb(st, tb, done):
    """Backward of the timestep path (sinusoid -> MLP -> all time_emb_proj), shared by both tapes (fp32 throughout)."""
    P = st.params
    dtproj, act = tb["dtproj"], tb["act"]
    for pre in tb["resnets"]:   # (resnets whose conv1 got no gradient this step)
        _temb_proj_grads(st, tb, pre, done)
    dact = ops.linear_bwd(act, tb["wp"], dtproj, None, None, need_dx=True)
    dz2 = ops.silu_bwd(tb["z2"], dact)
    # h1 = silu(z1) is recomputed by its definition through the kernels' own activation
    h1 = ops.silu_fwd(tb["z1"])
    dh1 = ops.linear_bwd(h1, tb["w2"], dz2, st.grad("time_embedding.linear_2.weight"),
                         st.grad("time_embedding.linear_2.bias"), need_dx=True)
    dz1 = ops.silu_bwd(tb["z1"], dh1)
    ops.linear_bwd(tb["emb"], P["time_embedding.linear_1.weight"].detach(), dz1,
                   st.grad("time_embedding.linear_1.weight"), st.grad("time_embedding.linear_1.bias"), need_dx=False)
    done("time_embedding.linear_2.weight", "time_embedding.linear_2.bias", "time_embedding.linear_1.weight",
         "time_embedding.linear_1.bias")


class _UNetTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, sample, timesteps):
        st = get_train_state(model)
        tape = _Tape()
        dt = _lib.DTYPE_CODES[getattr(model, "compute_dtype", "fp32")]
        if dt == _lib.DSG_F32:
            out = _forward(model, st, tape, sample, timesteps)
        else:
            out = _forward16(model, st, tape, sample, timesteps, dt)
        ctx.model, ctx.st, ctx.tape, ctx.out_ref, ctx.dt = model, st, tape, out, dt
        return out.view_as(out)  # a fresh tensor object for autograd; `out` keys the tape

    @staticmethod
    def backward(ctx, dout):
        st, tape = ctx.st, ctx.tape
        st.attach()
        dout = dout.contiguous()
        if ctx.dt != _lib.DSG_F32:
            # bf16 has fp32's exponent range; fp16 gets its range from the GradScaler (Accelerator), as in the reference
            tape.setg(ctx.out_ref, dout)
            _backward16(ctx.model, st, tape, dout)
            scale = 1.0
        else:
            # The gradient of a mean-reduced loss is O(1 / numel) per element (1e-7 at 256^2 x B = 16): below the range in
            # which the fp16 pairs of the split path keep fp32's precision.  The walk therefore runs on S * dout with
            # S = 2^floor(log2(numel)) -- activations' gradients are O(1) -- and the parameter gradients, which accumulate
            # in the slab, are brought back by 1 / S at the end: powers of two, exact, so what is in p.grad is what an
            # unscaled fp32 backward would have produced.  Gradients already in the slab (accumulation over
            # micro-batches) are taken along by the same factor first.
            scale = 2.0 ** math.floor(math.log2(max(1, dout.numel())))
            ops.scale(st.grad_flat, None, scale, out=st.grad_flat)
            tape.setg(ctx.out_ref, ops.scale(dout, None, scale))
            _backward(ctx.model, st, tape, dout)
        ctx.tape = None

        def finish():
            for hook in st.post_backward:
                hook()
            if scale != 1.0:
                ops.scale(st.grad_flat, None, 1.0 / scale, out=st.grad_flat)
        # after the whole autograd pass (this node is its last consumer of GPU work): finish collectives, remove the scale
        torch.autograd.Variable._execution_engine.queue_callback(finish)
        return None, None, None, None


def unet_forward_train(model, sample, timestep):
    b = sample.shape[0]
    t = model._timesteps_tensor(timestep, b, sample.device)
    anchor = getattr(model, "_grad_anchor", None)
    if anchor is None or anchor.device != sample.device:
        anchor = torch.zeros(1, device=sample.device, requires_grad=True)
        model._grad_anchor = anchor
    return _UNetTrainFn.apply(anchor, model, sample.contiguous(), t)


# ------------------------------------------------------------------------------------------------------------------
# Mixed-precision tape (Accelerator(mixed_precision='bf16' | 'fp16'); reference: train.py:24, training_pipeline.py:48-49,
# SURVEY App. A.6).  Same walk as above on the engine's 16-bit channel-blocked tensors [N, C/8, H, W, 8]: every conv /
# projection and its two gradients run once on the bf16 / f16 matrix cores with fp32 accumulation; GroupNorm statistics,
# the attention core (q, k, v, softmax: fp32 [N, C, L]), the loss, parameter gradients, master weights and the
# optimizer stay fp32 -- torch.autocast's split.  Shapes the 16-bit kernels do not take (conv_in / conv_out, pointwise
# and stride-2 weight gradients, maps under one tile) go through dsg_layout_convert_dt and the fp32 kernels.
# ------------------------------------------------------------------------------------------------------------------
class _Packs:
    """Per-parameter 16-bit operand images (dsg_conv_weight_pack), refreshed when the parameter changes."""

    def __init__(self, st, dt):
        self.st, self.dt, self.cache, self.sig = st, dt, {}, {}
        # every image / row copy made so far as a job of ops.PackTable: from the second step on, one launch at the start of the
        # forward refreshes all of them (refresh_all) instead of ~190 small ones on first use
        self.jobs, self.job_params, self.table = {}, {}, None

    def _sig(self, names):
        return tuple((self.st.params[n].data_ptr(), self.st.params[n]._version) for n in names)

    def _job(self, key, names, specs):
        """remember how `key` is refreshed: `specs` = PackTable jobs, `names` = the parameters whose change makes it stale"""
        if key not in self.jobs:
            self.table = None
        self.jobs[key], self.job_params[key] = specs, names

    def refresh_all(self):
        """One launch for every stale image (dsg_conv_weight_pack_batch) once the set of images is known -- same bits as the
        one-by-one path, which still serves anything new."""
        if len(self.jobs) < 16 or os.environ.get("DSG_NO_PACK_BATCH") == "1":
            return
        sigs = {key: self._sig(names) for key, names in self.job_params.items()}
        if all(self.sig.get(key) == sg for key, sg in sigs.items()):
            return
        # (a job names its parameter: the optimizer may have moved the parameters into its own slab since the job was recorded)
        flat = [dict(spec, w=self.st.params[spec["name"]].detach()) for specs in self.jobs.values() for spec in specs]
        ptrs = tuple((j["w"].data_ptr(), j["dst"].data_ptr()) for j in flat)
        if self.table is None or self.table.ptrs != ptrs:
            self.table = ops.PackTable(flat)
        self.table.run()
        self.sig.update(sigs)

    def get(self, name, kind):
        key = (name, kind)
        sig = self._sig((name,))
        if self.sig.get(key) != sig:
            p = self.st.params[name]
            self.cache[key] = ops.pack_conv_weight(p.detach(), kind, self.dt, out=self.cache.get(key))
            self.sig[key] = sig
            self._job(key, (name,), [dict(name=name, dst=self.cache[key], kind=kind, dtype=self.dt)])
        return self.cache[key]

    def copy_rows(self, key, name, dst):
        """dst <- the parameter's elements (rows of the fused time_emb_proj matrix), refreshed when it changes"""
        sig = self._sig((name,))
        if self.sig.get(key) != sig:
            p = self.st.params[name].detach()
            dst.copy_(p.reshape(dst.shape))
            self.sig[key] = sig
            self._job(key, (name,), [dict(name=name, dst=dst, kind=-1)])

    def qkv(self, prefix, kind):
        """Fused q/k/v projection: forward = three column windows of one [C] x [3C] image; data gradient = the three
        [C -> C] images back to back along K."""
        names = [f"{prefix}.{t}.weight" for t in ("to_q", "to_k", "to_v")]
        sig = self._sig(names)
        key = (prefix, "qkv", kind)
        if self.sig.get(key) != sig:
            ws = [self.st.params[n].detach() for n in names]
            c = ws[0].shape[0]
            if kind == ops.PACK_FWD:
                buf = self.cache.get(key)
                for i, w in enumerate(ws):
                    buf = ops.pack_conv_weight(w, ops.PACK_FWD, self.dt, n_total=3 * c, n_off=i * c, out=buf)
            else:
                one = ops.pack_conv_weight(ws[0], ops.PACK_DGRAD, self.dt)
                buf = self.cache.get(key)
                if buf is None:
                    buf = torch.zeros(3 * one.numel(), dtype=torch.int16, device=one.device)
                for i, w in enumerate(ws):
                    ops.pack_conv_weight(w, ops.PACK_DGRAD, self.dt, out=buf[i * one.numel():(i + 1) * one.numel()])
            self.cache[key] = buf
            self.sig[key] = sig
            if kind == ops.PACK_FWD:
                specs = [dict(name=n, dst=buf, kind=kind, dtype=self.dt, n_total=3 * c, n_off=i * c) for i, n in enumerate(names)]
            else:
                per = buf.numel() // 3
                specs = [dict(name=n, dst=buf[i * per:(i + 1) * per], kind=kind, dtype=self.dt) for i, n in enumerate(names)]
            self._job(key, tuple(names), specs)
        return self.cache[key]


def _pad64(c):
    return (c + 63) // 64 * 64


def _forward16(model, st: TrainState, tape: _Tape, sample, timesteps, dt):
    cfg = model.config
    P = st.params
    groups, eps = cfg.norm_num_groups, cfg.norm_eps
    if any(c % 8 for c in cfg.block_out_channels):
        raise NotImplementedError("mixed-precision training needs block_out_channels % 8 == 0 (channel-blocked tensors)")
    packs = st.packs16.setdefault(dt, _Packs(st, dt))
    tape.packs, tape.dt = packs, dt
    packs.refresh_all()

    resnets, toffs, proj_total = [], {}, 0
    for pre in _resnet_prefixes(model):
        toffs[pre] = proj_total
        proj_total += P[pre + ".time_emb_proj.bias"].numel()
        resnets.append(pre)

    # ---- timestep path (fp32, as in the fp32 tape) ----
    w1, b1 = P["time_embedding.linear_1.weight"].detach(), P["time_embedding.linear_1.bias"].detach()
    w2, b2 = P["time_embedding.linear_2.weight"].detach(), P["time_embedding.linear_2.bias"].detach()
    act, emb, z1, z2 = ops.time_embed_train(timesteps, w1, b1, w2, b2, st.freqs)
    dim = w1.shape[0]
    if "_tproj" not in st.wf:
        st.wf["_tproj"] = torch.empty((proj_total, dim), dtype=torch.float32, device=sample.device)
        st.wf["_tproj.bias"] = torch.empty(proj_total, dtype=torch.float32, device=sample.device)
    wp, bp = st.wf["_tproj"], st.wf["_tproj.bias"]
    for pre in resnets:
        o = toffs[pre]
        n = P[pre + ".time_emb_proj.bias"].numel()
        packs.copy_rows((pre, "tproj.w"), pre + ".time_emb_proj.weight", wp[o:o + n])
        packs.copy_rows((pre, "tproj.b"), pre + ".time_emb_proj.bias", bp[o:o + n])
    tproj = ops.linear(act, wp, bp)
    dtproj = torch.zeros_like(tproj)
    tape.temb = dict(act=act, emb=emb, z1=z1, z2=z2, wp=wp, dtproj=dtproj, toffs=toffs, resnets=resnets, w2=w2)

    pstats = {}

    def hw_of(x):
        return x.shape[2] * x.shape[3]

    def chans(x):
        return x.shape[1] * 8 if x.dim() == 5 else x.shape[1]

    def norm_ss(x0, x1, gn):
        g, b = P[gn + ".weight"].detach(), P[gn + ".bias"].detach()
        s0 = pstats.get(id(x0))
        if s0 is None:   # the producer could not write its statistics: one pass over the 16-bit tensor, kept with it
            s0 = pstats[id(x0)] = ops.gn_channel_stats_blocked(x0)
        s1 = None
        if x1 is not None:
            s1 = pstats.get(id(x1))
            if s1 is None:
                s1 = pstats[id(x1)] = ops.gn_channel_stats_blocked(x1)
        return ops.gn_scale_shift_from_parts_train(s0, g, b, groups, eps, hw_of(x0), stats1=s1)

    def conv(x0, wname, x1=None, gn=None, silu=False, k=3, stride=1, ups=False, toff=None, res=None, need_dx=True,
             feeds_norm=False, dst_blocked=True, ss_mr=None):
        bias = P[wname + ".bias"].detach()
        cout = bias.numel()
        src_blocked = x0.dim() == 5
        ss = mr = None
        if ss_mr is not None:     # (the caller has finalised this norm already: no second finalize launch)
            ss, mr = ss_mr
        elif gn is not None:
            ss, mr = norm_ss(x0, x1, gn)
        cin = chans(x0) + (chans(x1) if x1 is not None else 0)
        kw = {}
        if cin % 16 == 0 and cout % 8 == 0 and (src_blocked or dst_blocked):
            if stride == 2:
                kw["weight_h2_s2"] = packs.get(wname + ".weight", ops.PACK_S2)
            elif ups:
                kw["weight_h2_fold"] = packs.get(wname + ".weight", ops.PACK_FOLD)
            else:
                kw["weight_h2"] = packs.get(wname + ".weight", ops.PACK_FWD)
                kw["weight_h2_stride"] = _pad64(cout)
        y = st.lazy_w(wname + ".weight", "wf", lambda wf: ops.conv2d_fused(
            x0, wf, bias, src1=x1, ksize=k, stride=stride, upsample=ups, gn_scale_shift=ss, silu=silu,
            temb=None if toff is None else tproj[:, toff:], temb_stride=tproj.stride(0), residual=res,
            cout=cout, want_stats=feeds_norm, src_blocked=src_blocked, dst_blocked=dst_blocked, compute_dtype=dt, **kw))
        if feeds_norm:
            y, ystats = y
            if ystats is not None:
                pstats[id(y)] = ystats
        tape.recs.append(dict(kind="conv", x0=x0, x1=x1, ss=ss, mr=mr, gn=gn, silu=silu, k=k, stride=stride, ups=ups,
                              toff=toff, res=res, y=y, wname=wname, cout=cout, need_dx=need_dx))
        return y

    def resnet(x, skip, pre):
        h = conv(x, pre + ".conv1", x1=skip, gn=pre + ".norm1", silu=True, toff=toffs[pre], feeds_norm=True)
        if (pre + ".conv_shortcut.weight") not in P:
            return conv(h, pre + ".conv2", gn=pre + ".norm2", silu=True, res=x, feeds_norm=True)
        # conv_shortcut(input) + conv2(...): where the kernel takes it the 1x1 rides on conv2's K loop exactly as in the
        # inference plan (dsg_conv_args.sc_*: no shortcut tensor in HBM, no residual read); the tape still gets one record
        # per conv -- both see the resnet output's gradient, which is all either backward needs
        scn, c2n = pre + ".conv_shortcut", pre + ".conv2"
        cout = P[c2n + ".bias"].numel()
        cin_sc = chans(x) + (chans(skip) if skip is not None else 0)
        if x.dim() == 5 and cin_sc % 16 == 0 and cout % 8 == 0 and chans(h) % 16 == 0:
            ss, mr = norm_ss(h, None, pre + ".norm2")
            sc = dict(src0=x, src1=skip, weight_h2=packs.get(scn + ".weight", ops.PACK_FWD), bias=P[scn + ".bias"].detach())
            kw = dict(ksize=3, gn_scale_shift=ss, silu=True, cout=cout, src_blocked=True, dst_blocked=True, compute_dtype=dt,
                      weight_h2=packs.get(c2n + ".weight", ops.PACK_FWD), weight_h2_stride=_pad64(cout))
            b2 = P[c2n + ".bias"].detach()
            # (the answer depends on shapes, dtype and the kernel-selection switches only: asked once per layer and shape)
            fkey = (pre, tuple(h.shape), tuple(x.shape), None if skip is None else tuple(skip.shape), dt, ops.tuning_epoch())
            fuse = st.fuse_cache.get(fkey)
            if fuse is None:
                fuse = st.fuse_cache[fkey] = bool(ops.conv2d_fused(h, None, b2, shortcut=dict(sc, query_only=True), **kw))
            if fuse:
                y, ystats = ops.conv2d_fused(h, None, b2, shortcut=sc, want_stats=True, **kw)
                if ystats is not None:
                    pstats[id(y)] = ystats
                tape.recs.append(dict(kind="conv", x0=x, x1=skip, ss=None, mr=None, gn=None, silu=False, k=1, stride=1, ups=False,
                                      toff=None, res=None, y=y, wname=scn, cout=cout, need_dx=True))
                tape.recs.append(dict(kind="conv", x0=h, x1=None, ss=ss, mr=mr, gn=pre + ".norm2", silu=True, k=3, stride=1,
                                      ups=False, toff=None, res=None, y=y, wname=c2n, cout=cout, need_dx=True))
                return y
            sc = conv(x, scn, x1=skip, k=1)
            return conv(h, c2n, gn=pre + ".norm2", silu=True, res=sc, feeds_norm=True, ss_mr=(ss, mr))
        sc = conv(x, scn, x1=skip, k=1)
        return conv(h, c2n, gn=pre + ".norm2", silu=True, res=sc, feeds_norm=True)

    def attention(x, pre):
        wf, _, bias = st.qkv_w(pre)
        c = chans(x)
        heads = c // cfg.attention_head_dim
        gnn = pre + ".group_norm"
        ss, mr = norm_ss(x, None, gnn)
        n, _, hh, ww, _ = x.shape
        qkv = ops.conv2d_fused(x, wf, bias, ksize=1, gn_scale_shift=ss, silu=False, cout=3 * c, src_blocked=True,
                               dst_blocked=False, compute_dtype=dt, weight_h2=packs.qkv(pre, ops.PACK_FWD),
                               weight_h2_stride=_pad64(3 * c))
        o, lse = ops.attention_train(qkv.view(n, 3 * c, hh * ww), heads, dtype=dt)
        o = o.view(n, c, hh, ww)
        tape.recs.append(dict(kind="qkv", x=x, ss=ss, mr=mr, gn=gnn, pre=pre, qkv=qkv))
        tape.recs.append(dict(kind="attn", qkv=qkv, o=o, lse=lse, heads=heads))
        return conv(o, pre + ".to_out.0", k=1, res=x, feeds_norm=True)

    x = conv(sample, "conv_in", need_dx=False)
    skips = [x]
    for i, blk in enumerate(model.down_blocks):
        pre = f"down_blocks.{i}"
        for j in range(len(blk.resnets)):
            x = resnet(x, None, f"{pre}.resnets.{j}")
            if hasattr(blk, "attentions"):
                x = attention(x, f"{pre}.attentions.{j}")
            skips.append(x)
        if hasattr(blk, "downsamplers"):
            x = conv(x, f"{pre}.downsamplers.0.conv", stride=2, feeds_norm=True)
            skips.append(x)
    x = resnet(x, None, "mid_block.resnets.0")
    if hasattr(model.mid_block, "attentions"):
        x = attention(x, "mid_block.attentions.0")
    x = resnet(x, None, "mid_block.resnets.1")
    for i, blk in enumerate(model.up_blocks):
        pre = f"up_blocks.{i}"
        for j in range(len(blk.resnets)):
            s = skips.pop()
            x = resnet(x, s, f"{pre}.resnets.{j}")
            if hasattr(blk, "attentions"):
                x = attention(x, f"{pre}.attentions.{j}")
        if hasattr(blk, "upsamplers"):
            x = conv(x, f"{pre}.upsamplers.0.conv", ups=True, feeds_norm=True)
    return conv(x, "conv_out", gn="conv_norm_out", silu=True, dst_blocked=False)


def _resnet_prefixes(model):
    for i, blk in enumerate(model.down_blocks):
        for j in range(len(blk.resnets)):
            yield f"down_blocks.{i}.resnets.{j}"
    yield "mid_block.resnets.0"
    yield "mid_block.resnets.1"
    for i, blk in enumerate(model.up_blocks):
        for j in range(len(blk.resnets)):
            yield f"up_blocks.{i}.resnets.{j}"


def _backward16(model, st: TrainState, tape: _Tape, dout):
    P = st.params
    groups = model.config.norm_num_groups
    tb, packs, dt = tape.temb, tape.packs, tape.dt
    fire = st.grad_ready_hooks

    def done(*names):
        for h in fire:
            for nm in names:
                h(nm)

    def blocked(t):
        return t.dim() == 5

    def chans(t):
        return t.shape[1] * 8 if blocked(t) else t.shape[1]

    def f32(t):   # [N, C, H, W] fp32 view of a tape tensor (conversion pass for the shapes the 16-bit kernels skip)
        return ops.from_blocked(t) if blocked(t) else t

    def pad64(t, key):
        """a tape tensor (fp32 [N, C, H, W] or blocked 16-bit, C <= 64) as a blocked 16-bit tensor of 64 channels, the
        missing ones zero; the padded buffer is kept between steps (its zero channels are written once)"""
        if blocked(t) and t.shape[1] == 8:
            return t
        n = t.shape[0]
        h, w = (t.shape[2], t.shape[3])
        buf = st.scratch16.get((key, n, h, w, dt))
        if buf is None:
            buf = st.scratch16[(key, n, h, w, dt)] = torch.zeros((n, 8, h, w, 8), dtype=_lib.TORCH_DTYPES[ops.dtype_code(dt)],
                                                                 device=t.device)
        if blocked(t):
            buf[:, :t.shape[1]].copy_(t)
        else:
            c = t.shape[1]
            if c % 8:
                x8 = st.scratch16.get((key, "nchw", n, h, w))
                if x8 is None:
                    x8 = st.scratch16[(key, "nchw", n, h, w)] = torch.zeros((n, (c + 7) // 8 * 8, h, w), dtype=torch.float32,
                                                                            device=t.device)
                x8[:, :c].copy_(t)
                t = x8
            buf[:, :t.shape[1] // 8].copy_(ops.to_blocked(t.contiguous(), dt))
        return buf

    def pad_ss(ss, c):
        """[N, c, 2] scale / shift -> [N, 64, 2] (the padded channels are zero anyway)"""
        if ss is None or c == 64:
            return ss
        out = torch.zeros((ss.shape[0], 64, 2), dtype=ss.dtype, device=ss.device)
        out[:, :c].copy_(ss)
        return out

    def wgrad(x0, x1, dy, wname, k, stride, ups, ss, silu, cout=None, dy_coff=0, sums=None, sums_stride=0, bias_grad=None):
        """weight gradient; returns True when `sums` (per-(n, cout) sums of dy) was filled as a by-product -- and, with
        `bias_grad`, 2 when their sum over the batch was added to it in the same pass"""
        c0, c1 = chans(x0), (chans(x1) if x1 is not None else 0)
        co = cout or chans(dy)
        if (blocked(dy) and (blocked(x0) or x1 is None)
                and ops.wgrad16_supported(c0, c1, co, x0.shape[2], x0.shape[3], k, stride, ups, dy_coff)):
            # (an fp32 [N, C, H, W] source -- the attention output in front of to_out -- is rounded to the tape's type first)
            # (the sampler convs come through here too: Upsample2D's conv with x at HALF the K grid's resolution, the stride-2 conv
            #  with dY at half of it -- addressed through a shift in the kernel, no x2 copy of x, no zero-stuffed copy of dY)
            xb = x0 if blocked(x0) else ops.to_blocked(x0.contiguous(), dt)
            ops.conv_wgrad(xb, dy, st.grad(wname), src1=x1, ksize=k, stride=stride, upsample=ups, gn_scale_shift=ss, silu=silu,
                           cout=co, dy_coff=dy_coff, dy_sums=sums, dy_sums_stride=sums_stride,
                           bias_grad=bias_grad if sums is not None else None)
            return (2 if bias_grad is not None else True) if sums is not None else False
        elif (k == 3 and stride == 2 and not ups and x1 is None and dy_coff == 0 and blocked(x0) and blocked(dy) and ss is None
              and ops.wgrad16_supported(c0, 0, co, x0.shape[2], x0.shape[3], 3)):
            # down-sampler conv: its weight gradient is the stride-1 one against dY with zeros between its pixels
            # (dY_up[2y, 2x] = dY[y, x]): the 16-bit kernel at full resolution instead of two conversions to fp32 and the
            # exact f32 kernel.  The zero positions of the buffer are written once and kept between steps.
            key = ("dyup", dt, wname, tuple(dy.shape))
            up = st.scratch16.get(key)
            if up is None:
                up = st.scratch16[key] = torch.zeros((dy.shape[0], dy.shape[1], x0.shape[2], x0.shape[3], 8), dtype=dy.dtype,
                                                     device=dy.device)
            up[:, :, ::2, ::2].copy_(dy)
            ops.conv_wgrad(x0, up, st.grad(wname), ksize=3, cout=co, dy_sums=sums, dy_sums_stride=sums_stride)
            return sums is not None
        elif (k == 3 and stride == 1 and not ups and x1 is None and dy_coff == 0 and c0 <= 64 and co <= 64
              and ops.wgrad16_supported(64, 0, 64, x0.shape[2], x0.shape[3], 3)):
            # conv_in (few input channels) / conv_out (few output channels): the 16-bit kernel wants 64-channel blocks, so
            # the narrow side is padded with zero channels -- a 64 x 64 weight gradient of which one corner is kept.  (The
            # fp32 detour -- dY or x converted back to [N,C,H,W], the exact f32 kernel on a 2 M-pixel contraction with 8
            # rows -- was 2.3 ms of the bf16 step; this is 0.6.)
            xb, dyb = pad64(x0, "x:" + wname), pad64(dy, "dy:" + wname)
            tmp = st.scratch16.get(("dw", wname))
            if tmp is None:
                tmp = st.scratch16[("dw", wname)] = torch.empty((64, 64, 3, 3), dtype=torch.float32, device=dy.device)
            tmp.zero_()
            ops.conv_wgrad(xb, dyb, tmp, ksize=3, gn_scale_shift=pad_ss(ss, c0), silu=silu, cout=64)
            g = st.grad(wname)
            g.copy_(ops.add(g, tmp[:co, :c0].contiguous()))
            return False
        else:
            ops.conv_wgrad(f32(x0), f32(dy), st.grad(wname), src1=None if x1 is None else f32(x1), ksize=k, stride=stride,
                           upsample=ups, gn_scale_shift=ss, silu=silu, cout=co, dy_coff=dy_coff)
            return False

    def dgrad(dy, wname, k, cout, stride=1, residual=None, col0=0, ncols=None, dst_blocked=True, gnb=None):
        """dX = conv(dY, W^T flipped) for columns [col0, col0 + ncols) of the conv's input channels; + residual.
        gnb (dict(x0=, x1=, ss=, silu=)): the conv sits behind a GroupNorm -- returns (dX, parts): parts = the norm's backward
        statistics from the kernel's epilogue (dsg_conv_args.gnb_*) where the call's kernel has that form, else None."""
        wd_stride = ops._pad32(cout) + 64   # row length of the fp32 data-gradient layout (st.wd_of)
        ncols = ncols or cout
        full = col0 == 0 and ncols == cout
        src_blocked = blocked(dy)
        kdim = chans(dy)
        use16 = kdim % 16 == 0 and ncols % 8 == 0 and (src_blocked or dst_blocked) and (full or (col0 % 8 == 0 and ncols % 64 == 0))
        if stride == 2 and gnb is not None:
            return dgrad(dy, wname, k, cout, stride, residual, col0, ncols, dst_blocked), None
        if stride == 2:   # the adjoint of the space-to-depth conv: four 2x2 phase convs of the low-resolution dY
            ok = use16 and full and src_blocked and dst_blocked and dy.shape[3] % 32 == 0 and dy.shape[2] % 8 == 0
            if ok:
                return st.lazy_w(wname, "wd", lambda wd: ops.conv2d_fused(
                    dy, wd, ksize=3, upsample=True, cout=cout, residual=residual, src_blocked=True, dst_blocked=True,
                    compute_dtype=dt, weight_h2_fold=packs.get(wname, ops.PACK_DGRAD_S2)))
            g = ops.conv2d_fused(f32(dy), st.wd_of(wname), ksize=3, upsample=2, cout=cout,
                                 residual=None if residual is None else f32(residual))
            return ops.to_blocked(g, dt)
        if use16:
            def run16(wd):
                kw = dict(ksize=k, cout=ncols, residual=residual, wstride=None if full else wd_stride, src_blocked=src_blocked,
                          dst_blocked=dst_blocked, compute_dtype=dt, weight_h2=packs.get(wname, ops.PACK_DGRAD),
                          weight_h2_col=col0, weight_h2_stride=_pad64(cout))
                w = wd if (full or wd is None) else wd[:, :, col0:]
                if gnb is None:
                    return ops.conv2d_fused(dy, w, **kw)
                if full and residual is None and ops.conv2d_fused(dy, w, gnb=dict(gnb, query_only=True), **kw):
                    return ops.conv2d_fused(dy, w, gnb=gnb, want_stats=True, **kw)
                return ops.conv2d_fused(dy, w, **kw), None
            return st.lazy_w(wname, "wd", run16)
        if gnb is not None:   # (the fp32 detours below have no such epilogue)
            return dgrad(dy, wname, k, cout, stride, residual, col0, ncols, dst_blocked), None
        # no 16-bit kernel for this shape (conv_out's 8-channel dY, narrow channel windows): the fp32 kernels
        wd = st.wd_of(wname)
        if not src_blocked and dst_blocked and full and k == 3:   # fp32 [N,C,H,W] dY -> 16-bit blocked dX directly
            return ops.conv2d_fused(dy, wd, ksize=k, cout=cout, residual=residual, dst_blocked=True, compute_dtype=dt)
        g = ops.conv2d_fused(f32(dy), wd if full else wd[:, :, col0:], ksize=k, cout=ncols,
                             residual=None if residual is None else f32(residual), wstride=None if full else wd_stride)
        return ops.to_blocked(g, dt) if dst_blocked else g

    for rec in reversed(tape.recs):
        kind = rec["kind"]
        if kind == "conv":
            dy = tape.g(rec["y"])
            if dy is None:
                continue
            wname, cout, k = rec["wname"], rec["cout"], rec["k"]
            x0, x1 = rec["x0"], rec["x1"]
            # per-(n, c) sums of dy (bias and time-embedding gradients): a by-product of the 16-bit weight-gradient
            # kernel where that serves the conv, a pass of their own otherwise
            if rec["toff"] is not None:
                sums, sstride = tb["dtproj"][:, rec["toff"]:], tb["dtproj"].stride(0)
            else:
                sums = torch.empty((dy.shape[0], cout), dtype=torch.float32, device=dy.device)
                sstride = cout
            if rec["res"] is not None:
                tape.addg(rec["res"], dy, lazy=True)
            bg = st.grad(wname + ".bias")
            if rec["ups"]:   # weight gradient against the nearest-x2 input, data gradient at full resolution
                if x1 is None and blocked(x0) and blocked(dy) and ops.wgrad16_supported(chans(x0), 0, cout, x0.shape[2], x0.shape[3],
                                                                                       k, 1, True):
                    have = wgrad(x0, None, dy, wname + ".weight", k, 1, True, None, False,     # (x read at (y >> 1, x >> 1))
                                 sums=sums, sums_stride=sstride, bias_grad=bg)
                else:   # (shapes the kernel's sampler form does not take: the materialised x2 input)
                    have = wgrad(ops.upsample_nearest2x(x0), None, dy, wname + ".weight", k, 1, False, None, False,
                                 sums=sums, sums_stride=sstride, bias_grad=bg)
            else:
                have = wgrad(x0, x1, dy, wname + ".weight", k, rec["stride"], False, rec["ss"], rec["silu"],
                             sums=sums, sums_stride=sstride, bias_grad=bg)
            if not have:
                ops.channel_sums(dy, out=sums, out_stride=sstride)
            if have != 2:   # (2: the weight-gradient call finished the bias gradient with the sums)
                ops.reduce_rows_add(sums, bg, stride=sstride)
            done(wname + ".weight", wname + ".bias")
            if rec["toff"] is not None:
                _temb_proj_grads(st, tb, wname[:-len(".conv1")], done)
            if not rec["need_dx"]:
                continue
            cin0, cin1 = chans(x0), (chans(x1) if x1 is not None else 0)
            wn = wname + ".weight"
            if rec["ups"]:
                # one 4x4 stride-2 window over dY per low-resolution pixel (16 taps; dsg_conv_args.s2_window4) where the
                # space-to-depth kernel takes the shape; else the 3x3 data gradient at full resolution + 2x2 sums (36 taps)
                if (not _UPS_DGRAD_FULLRES and k == 3 and blocked(dy) and blocked(x0) and cout % 8 == 0 and cin0 % 8 == 0
                        and x0.shape[2] % 8 == 0 and (x0.shape[3] % 32 == 0 or x0.shape[3] in (8, 16))):
                    tape.setg(x0, ops.conv2d_fused(dy, None, ksize=3, stride=2, cout=cin0, residual=tape.g(x0), src_blocked=True,
                                                   dst_blocked=True, compute_dtype=dt, s2_window4=True,
                                                   weight_h2_s2=packs.get(wn, ops.PACK_DGRAD_UPS)))
                    continue
                dfull = dgrad(dy, wn, k, cin0)
                tape.setg(x0, ops.sumpool2x2(dfull, add=tape.g(x0)))
                continue
            if rec["gn"] is not None:
                da, parts = dgrad(dy, wn, k, cin0 + cin1, stride=rec["stride"],
                                  gnb=dict(x0=x0, x1=x1, ss=rec["ss"], silu=rec["silu"]))
                gnn = rec["gn"]
                a0, a0b = tape.g2(x0)
                dx0, dx1 = ops.gn_bwd_blocked(x0, da, rec["ss"], rec["mr"], P[gnn + ".weight"].detach(), groups, rec["silu"],
                                              st.grad(gnn + ".weight"), st.grad(gnn + ".bias"), src1=x1, add0=a0, add0b=a0b,
                                              add1=tape.g(x1) if x1 is not None else None, parts=parts)
                done(gnn + ".weight", gnn + ".bias")
                tape.setg(x0, dx0)
                if x1 is not None:
                    tape.setg(x1, dx1)
            elif x1 is None:
                tape.setg(x0, dgrad(dy, wn, k, cin0, stride=rec["stride"], residual=tape.g(x0), dst_blocked=blocked(x0)))
            elif cin0 % 8 == 0 and cin1 % 64 == 0:   # one conv per source on that source's columns of W^T
                tape.setg(x0, dgrad(dy, wn, k, cin0 + cin1, residual=tape.g(x0), col0=0, ncols=cin0))
                tape.setg(x1, dgrad(dy, wn, k, cin0 + cin1, residual=tape.g(x1), col0=cin0, ncols=cin1))
            else:   # narrow second source: one conv over the concatenation, split by channel block
                da = dgrad(dy, wn, k, cin0 + cin1)
                tape.addg(x0, da[:, :cin0 // 8].contiguous())
                tape.addg(x1, da[:, cin0 // 8:].contiguous())
        elif kind == "attn":
            do = tape.g(rec["o"])
            qkv = rec["qkv"]
            n, c3, hh, ww = qkv.shape
            # (the attention core's backward on the matrix cores, operands rounded once to the tape's 16-bit type --
            #  autocast's split; in fp16 dO is scaled by a power of two inside the kernels, so dS stays in fp16's range)
            dqkv = ops.attention_bwd(qkv.view(n, c3, hh * ww), rec["o"].view(n, c3 // 3, hh * ww),
                                     do.view(n, c3 // 3, hh * ww), rec["lse"], rec["heads"],
                                     dtype=dt)
            tape.setg(qkv, dqkv.view(n, c3, hh, ww))
        elif kind == "qkv":
            dqkv = tape.g(rec["qkv"])
            x, pre = rec["x"], rec["pre"]
            c = chans(x)
            sums = ops.channel_sums(dqkv)
            # the three projections' weight gradients on the 16-bit kernel: dqkv (fp32 [N, 3C, L], the attention
            # backward's result) is rounded to the tape's type once; shapes the kernel does not take keep the fp32 form
            # (bf16 only: dqkv's entries for K are small enough to lose bits in fp16's range even under the loss scale --
            # the worst tensor of the fp16 parity test sat at its tolerance)
            w16 = dt == _lib.DSG_BF16 and ops.wgrad16_supported(c, 0, c, x.shape[2], x.shape[3], 1, 1, False, 0)
            xw, dyw = (x, ops.to_blocked(dqkv.contiguous(), dt)) if w16 else (ops.from_blocked(x), dqkv)
            for i, t in enumerate(("to_q", "to_k", "to_v")):
                ops.reduce_rows_add(sums[:, i * c:], st.grad(f"{pre}.{t}.bias"), stride=sums.stride(0))
                ops.conv_wgrad(xw, dyw, st.grad(f"{pre}.{t}.weight"), ksize=1, gn_scale_shift=rec["ss"], silu=False,
                               cout=c, dy_coff=i * c)
                done(f"{pre}.{t}.weight", f"{pre}.{t}.bias")
            _, wd, _ = st.qkv_w(pre)
            if c % 64 == 0:
                da = ops.conv2d_fused(dqkv, wd, ksize=1, cout=c, src_blocked=False, dst_blocked=True, compute_dtype=dt,
                                      weight_h2=packs.qkv(pre, ops.PACK_DGRAD), weight_h2_stride=_pad64(c))
            else:
                da = ops.to_blocked(ops.conv2d_fused(dqkv, wd, ksize=1, cout=c), dt)
            gnn = rec["gn"]
            dx, _ = ops.gn_bwd_blocked(x, da, rec["ss"], rec["mr"], P[gnn + ".weight"].detach(), groups, False,
                                       st.grad(gnn + ".weight"), st.grad(gnn + ".bias"), add0=tape.g(x))
            done(gnn + ".weight", gnn + ".bias")
            tape.setg(x, dx)
    _backward_temb(st, tb, done)
