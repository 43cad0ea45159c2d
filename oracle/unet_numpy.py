"""A SECOND, independent restatement of diffusers-0.20.0 ``UNet2DModel.forward`` -- TEST INFRASTRUCTURE (oracle cross-check).

``oracle/unet_oracle.py`` composes torch's own ``nn.Conv2d`` / ``nn.GroupNorm`` / ``F.scaled_dot_product_attention`` the way
diffusers does.  This file shares nothing with it: fp64 numpy, no module tree -- a function that walks a STATE DICT by the
diffusers key names and writes every operator out from its definition (the
configuration of /root/reference/DriveSceneGen/scripts/train.py:39-57):

  conv2d        zero padding, cross-correlation  y[o, i, j] = b[o] + sum_{c, u, v} w[o, c, u, v] x[c, s i + u - p, s j + v - p]
  group norm    per (sample, group) mean and BIASED variance over (channels of the group, H, W), eps inside the square root
  attention     per head softmax(q k^T / sqrt(d)) v with d = attention_head_dim, heads taken as CONSECUTIVE channel runs
  upsample      nearest x2:  y[2i + a, 2j + b] = x[i, j]
  timestep emb  [cos | sin] of  t * exp(-ln(10000) * k / half)   (flip_sin_to_cos=True, freq_shift=0)

It pins the torch oracle against slips in how it USES torch (GroupNorm's eps / variance convention, SDPA's scale and head
layout, Conv2d's padding and stride, the skip bookkeeping) -- not against diffusers itself, which cannot be imported here
(parity unpinned, see oracle/__init__.py).  Small networks only (pure numpy convs): tests/test_oracle_kat.py runs it on
BASELINE configs[0] and on the attention-block network.
"""
from __future__ import annotations

import math

import numpy as np


def _silu(x):
    return x / (1.0 + np.exp(-x))


def _conv2d(x, w, b, stride=1):
    """x [N, C, H, W], w [O, C, k, k] (k = 3 with padding 1, or k = 1), b [O]"""
    n, c, h, wd = x.shape
    o, _, k, _ = w.shape
    p = k // 2
    xp = np.zeros((n, c, h + 2 * p, wd + 2 * p))
    xp[:, :, p:p + h, p:p + wd] = x
    ho, wo = (h + 2 * p - k) // stride + 1, (wd + 2 * p - k) // stride + 1
    y = np.zeros((n, o, ho, wo))
    for u in range(k):
        for v in range(k):
            patch = xp[:, :, u:u + stride * (ho - 1) + 1:stride, v:v + stride * (wo - 1) + 1:stride]   # x[c, s i + u - p, s j + v - p]
            y += np.einsum("oc,nchw->nohw", w[:, :, u, v], patch)
    return y + b[None, :, None, None]


def _group_norm(x, gamma, beta, groups, eps):
    n, c = x.shape[:2]
    xg = x.reshape(n, groups, -1)
    mean = xg.mean(axis=2, keepdims=True)
    var = ((xg - mean) ** 2).mean(axis=2, keepdims=True)          # biased
    xn = ((xg - mean) / np.sqrt(var + eps)).reshape(x.shape)
    shape = (1, c) + (1,) * (x.ndim - 2)
    return xn * gamma.reshape(shape) + beta.reshape(shape)


def _linear(x, w, b):
    return x @ w.T + b


def timestep_embedding(t, dim):
    """fp32, like diffusers' get_timestep_embedding (the embedding is built in float32 whatever the model's dtype)"""
    half = dim // 2
    freqs = np.exp((np.float32(-math.log(10000.0)) * np.arange(half, dtype=np.float32)) / np.float32(half))
    a = t.astype(np.float32)[:, None] * freqs[None, :]
    return np.concatenate([np.cos(a), np.sin(a)], axis=1).astype(np.float64)


class _Net:
    """the forward as a walk over state-dict keys"""

    def __init__(self, cfg, sd):
        self.cfg = cfg
        self.sd = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items()}
        self.groups = cfg.get("norm_num_groups", 32)
        self.eps = cfg.get("norm_eps", 1e-5)
        self.head_dim = cfg.get("attention_head_dim", 8)

    def p(self, key):
        return self.sd[key]

    def has(self, key):
        return key in self.sd

    def conv(self, pre, x, stride=1):
        return _conv2d(x, self.p(pre + ".weight"), self.p(pre + ".bias"), stride)

    def norm(self, pre, x):
        return _group_norm(x, self.p(pre + ".weight"), self.p(pre + ".bias"), self.groups, self.eps)

    def resnet(self, pre, x, temb):
        h = self.conv(pre + ".conv1", _silu(self.norm(pre + ".norm1", x)))
        h = h + _linear(_silu(temb), self.p(pre + ".time_emb_proj.weight"), self.p(pre + ".time_emb_proj.bias"))[:, :, None, None]
        h = self.conv(pre + ".conv2", _silu(self.norm(pre + ".norm2", h)))
        if self.has(pre + ".conv_shortcut.weight"):
            x = self.conv(pre + ".conv_shortcut", x)
        return x + h

    def attention(self, pre, x):
        n, c, hh, ww = x.shape
        tokens = self.norm(pre + ".group_norm", x).reshape(n, c, hh * ww).transpose(0, 2, 1)      # [N, L, C]
        q = _linear(tokens, self.p(pre + ".to_q.weight"), self.p(pre + ".to_q.bias"))
        k = _linear(tokens, self.p(pre + ".to_k.weight"), self.p(pre + ".to_k.bias"))
        v = _linear(tokens, self.p(pre + ".to_v.weight"), self.p(pre + ".to_v.bias"))
        d = self.head_dim
        out = np.zeros_like(q)
        for hd in range(c // d):                       # head = a run of d consecutive channels
            sl = slice(hd * d, (hd + 1) * d)
            s = q[:, :, sl] @ k[:, :, sl].transpose(0, 2, 1) / math.sqrt(d)
            s = s - s.max(axis=2, keepdims=True)
            pr = np.exp(s)
            pr = pr / pr.sum(axis=2, keepdims=True)
            out[:, :, sl] = pr @ v[:, :, sl]
        out = _linear(out, self.p(pre + ".to_out.0.weight"), self.p(pre + ".to_out.0.bias"))
        return out.transpose(0, 2, 1).reshape(n, c, hh, ww) + x

    def forward(self, sample, t):
        cfg = self.cfg
        boc = tuple(cfg["block_out_channels"])
        lpb = cfg.get("layers_per_block", 2)
        x = np.asarray(sample, dtype=np.float64)
        t = np.broadcast_to(np.asarray(t).reshape(-1), (x.shape[0],))
        temb = timestep_embedding(t, boc[0])
        temb = _linear(_silu(_linear(temb, self.p("time_embedding.linear_1.weight"), self.p("time_embedding.linear_1.bias"))),
                       self.p("time_embedding.linear_2.weight"), self.p("time_embedding.linear_2.bias"))
        x = self.conv("conv_in", x)
        skips = [x]
        for i, kind in enumerate(cfg["down_block_types"]):
            for j in range(lpb):
                x = self.resnet(f"down_blocks.{i}.resnets.{j}", x, temb)
                if kind == "AttnDownBlock2D":
                    x = self.attention(f"down_blocks.{i}.attentions.{j}", x)
                skips.append(x)
            if i != len(boc) - 1:
                x = self.conv(f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
                skips.append(x)
        x = self.resnet("mid_block.resnets.0", x, temb)
        if cfg.get("add_attention", True):
            x = self.attention("mid_block.attentions.0", x)
        x = self.resnet("mid_block.resnets.1", x, temb)
        for i, kind in enumerate(cfg["up_block_types"]):
            for j in range(lpb + 1):
                x = np.concatenate([x, skips.pop()], axis=1)
                x = self.resnet(f"up_blocks.{i}.resnets.{j}", x, temb)
                if kind == "AttnUpBlock2D":
                    x = self.attention(f"up_blocks.{i}.attentions.{j}", x)
            if i != len(boc) - 1:
                x = np.repeat(np.repeat(x, 2, axis=2), 2, axis=3)
                x = self.conv(f"up_blocks.{i}.upsamplers.0.conv", x)
        assert not skips
        return self.conv("conv_out", _silu(self.norm("conv_norm_out", x)))


def unet_forward(cfg: dict, state_dict: dict, sample, timestep):
    """eps = UNet2DModel(**cfg)(sample, timestep).sample in fp64 numpy; state_dict: diffusers keys -> arrays."""
    return _Net(cfg, state_dict).forward(sample, timestep)
