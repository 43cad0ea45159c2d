"""Torch-CPU restatement of diffusers-0.20.0 ``UNet2DModel`` -- TEST INFRASTRUCTURE (oracle).

Follows the configuration at /root/reference/DriveSceneGen/scripts/train.py:39-57 and the
semantics written down in SURVEY.md Appendix A.1/A.2 (diffusers 0.20.0 is pinned at
/root/reference/requirements.txt:15 but is not vendored; parity unpinned, see oracle/__init__.py).

Built only from the torch primitives diffusers itself composes: ``nn.Conv2d``, ``nn.GroupNorm``,
``nn.Linear``, ``F.silu``, ``F.scaled_dot_product_attention``, ``F.interpolate(nearest)``,
``torch.cat``.  Parameter names equal the diffusers state-dict keys (Appendix A.5).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


def timestep_embedding(timesteps: torch.Tensor, dim: int, flip_sin_to_cos: bool = True,
                       freq_shift: float = 0.0) -> torch.Tensor:
    """diffusers ``get_timestep_embedding`` (Appendix A.2 line 2): cos first when flipped."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_ch: int, dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_ch, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    """Appendix A.2 ``ResnetBlock2D(x, temb)``; used by train.py:39-57 through Down/UpBlock2D."""

    def __init__(self, in_ch: int, out_ch: int, temb_ch: int, groups: int, eps: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_ch, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_ch, out_ch, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, out_ch)
        self.norm2 = nn.GroupNorm(groups, out_ch, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_ch, out_ch, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_ch, out_ch, 1) if in_ch != out_ch else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class _ToOut(nn.ModuleList):
    pass


class Attention(nn.Module):
    """Appendix A.2 ``Attention(x)`` (deprecated-attn-block form, residual, GN without SiLU)."""

    def __init__(self, ch: int, head_dim: int, groups: int, eps: float):
        super().__init__()
        self.heads = ch // head_dim
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps, affine=True)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = _ToOut([nn.Linear(ch, ch), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, hh, ww = x.shape
        r = x
        h = x.view(b, c, hh * ww).transpose(1, 2)
        h = self.group_norm(h.transpose(1, 2)).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        d = c // self.heads
        q = q.view(b, -1, self.heads, d).transpose(1, 2)
        k = k.view(b, -1, self.heads, d).transpose(1, 2)
        v = v.view(b, -1, self.heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, c)
        o = self.to_out[1](self.to_out[0](o))
        o = o.transpose(-1, -2).reshape(b, c, hh, ww)
        return o + r


class Downsample2D(nn.Module):
    def __init__(self, ch: int, padding: int = 1):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=padding)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch: int):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    """DownBlock2D / AttnDownBlock2D (Appendix A.1, A.2)."""

    def __init__(self, in_ch, out_ch, temb_ch, num_layers, groups, eps, add_downsample, attn, head_dim):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, temb_ch, groups, eps) for i in range(num_layers)])
        if attn:
            self.attentions = nn.ModuleList([Attention(out_ch, head_dim, groups, eps) for _ in range(num_layers)])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch)]) if add_downsample else None

    def forward(self, x, temb):
        outs = ()
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class UpBlock(nn.Module):
    """UpBlock2D / AttnUpBlock2D (Appendix A.1, A.2)."""

    def __init__(self, in_ch, prev_ch, out_ch, temb_ch, num_layers, groups, eps, add_upsample, attn, head_dim):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip = in_ch if i == num_layers - 1 else out_ch
            rin = prev_ch if i == 0 else out_ch
            res.append(ResnetBlock2D(rin + skip, out_ch, temb_ch, groups, eps))
        self.resnets = nn.ModuleList(res)
        if attn:
            self.attentions = nn.ModuleList([Attention(out_ch, head_dim, groups, eps) for _ in range(num_layers)])
        else:
            self.attentions = None
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if add_upsample else None

    def forward(self, x, skips, temb):
        for i, res in enumerate(self.resnets):
            s = skips[-1]
            skips = skips[:-1]
            x = torch.cat([x, s], dim=1)
            x = res(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class MidBlock(nn.Module):
    def __init__(self, ch, temb_ch, groups, eps, add_attention, head_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_ch, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Attention(ch, head_dim, groups, eps)]) if add_attention else None

    def forward(self, x, temb):
        x = self.resnets[0](x, temb)
        if self.attentions is not None:
            x = self.attentions[0](x)
        return self.resnets[1](x, temb)


class OracleUNet2DModel(nn.Module):
    """``UNet2DModel`` as constructed at /root/reference/DriveSceneGen/scripts/train.py:39-57."""

    def __init__(self, sample_size=None, in_channels=3, out_channels=3,
                 down_block_types=("DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"),
                 up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
                 block_out_channels=(224, 448, 672, 896), layers_per_block=2, norm_num_groups=32,
                 norm_eps=1e-5, attention_head_dim=8, add_attention=True):
        super().__init__()
        boc = tuple(block_out_channels)
        self.config = SimpleNamespace(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=boc, layers_per_block=layers_per_block, norm_num_groups=norm_num_groups,
            norm_eps=norm_eps, attention_head_dim=attention_head_dim, add_attention=add_attention)
        temb_ch = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb_ch)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, t in enumerate(down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            final = i == len(boc) - 1
            hd = attention_head_dim if attention_head_dim is not None else out_ch
            self.down_blocks.append(DownBlock(in_ch, out_ch, temb_ch, layers_per_block, norm_num_groups, norm_eps,
                                              not final, t == "AttnDownBlock2D", hd))
        hd = attention_head_dim if attention_head_dim is not None else boc[-1]
        self.mid_block = MidBlock(boc[-1], temb_ch, norm_num_groups, norm_eps, add_attention, hd)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i, t in enumerate(up_block_types):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            final = i == len(boc) - 1
            hd = attention_head_dim if attention_head_dim is not None else out_ch
            self.up_blocks.append(UpBlock(in_ch, prev, out_ch, temb_ch, layers_per_block + 1, norm_num_groups,
                                          norm_eps, not final, t == "AttnUpBlock2D", hd))
        g_out = norm_num_groups if norm_num_groups is not None else min(boc[0] // 4, 32)
        self.conv_norm_out = nn.GroupNorm(g_out, boc[0], eps=norm_eps)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def forward(self, sample, timestep, return_dict: bool = True):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        elif t.dim() == 0:
            t = t[None].to(sample.device)
        t = t * torch.ones(sample.shape[0], dtype=t.dtype, device=t.device)
        t_emb = timestep_embedding(t, self.config.block_out_channels[0]).to(self.dtype)
        temb = self.time_embedding(t_emb)
        x = self.conv_in(sample)
        skips = (x,)
        for blk in self.down_blocks:
            x, outs = blk(x, temb)
            skips += outs
        x = self.mid_block(x, temb)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            s, skips = skips[-n:], skips[:-n]
            x = blk(x, s, temb)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        if not return_dict:
            return (x,)
        return SimpleNamespace(sample=x)
