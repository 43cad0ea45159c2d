"""``UNet2DModel`` -- the object DriveSceneGen's entry points hand around, backed by libdsg.so.

Mirrors the diffusers-0.20.0 class as the reference uses it:
 - constructor kwargs of /root/reference/DriveSceneGen/scripts/train.py:39-57 (+ diffusers defaults),
 - ``model(x, timesteps, return_dict=False)[0]`` (training_pipeline.py:84) and ``unet(image, t).sample``
   with a scalar ``t`` (the DDPMPipeline loop), ``unet.config.in_channels / sample_size``,
   ``unet.device / dtype``, ``parameters()`` (train.py:60,66), ``from_pretrained(dir, subfolder="unet")``
   (train.py:59), the App. A.5 state-dict keys and checkpoint folder.

The ``nn.Conv2d / nn.GroupNorm / nn.Linear`` sub-modules are parameter containers only (names, shapes,
default inits, ``state_dict``); their ``forward`` is never used -- every FLOP runs in the HIP engine,
and a CPU tensor raises (no fallback).
"""
from __future__ import annotations

import ctypes as C
import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib

_DEFAULTS = dict(
    sample_size=None, in_channels=3, out_channels=3, center_input_sample=False,
    time_embedding_type="positional", freq_shift=0, flip_sin_to_cos=True,
    down_block_types=("DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"),
    up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
    block_out_channels=(224, 448, 672, 896), layers_per_block=2, mid_block_scale_factor=1,
    downsample_padding=1, downsample_type="conv", upsample_type="conv", act_fn="silu",
    attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5, resnet_time_scale_shift="default",
    add_attention=True, class_embed_type=None, num_class_embeds=None)


class FrozenConfig(SimpleNamespace):
    """Attribute + mapping access, like diffusers' FrozenDict."""

    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, default=None):
        return getattr(self, k, default)

    def to_dict(self):
        return dict(self.__dict__)


class _Resnet(nn.Module):
    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)


class _Attention(nn.Module):
    def __init__(self, ch, groups, eps):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])


class _Sampler(nn.Module):
    def __init__(self, ch, stride):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=stride, padding=1)


class _Block(nn.Module):
    def __init__(self, resnets, attentions, sampler, sampler_name):
        super().__init__()
        self.resnets = nn.ModuleList(resnets)
        if attentions is not None:
            self.attentions = nn.ModuleList(attentions)
        if sampler is not None:
            setattr(self, sampler_name, nn.ModuleList([sampler]))


class _TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)


class UNet2DOutput(SimpleNamespace):
    pass


_SUPPORTED_DOWN = ("DownBlock2D", "AttnDownBlock2D")
_SUPPORTED_UP = ("UpBlock2D", "AttnUpBlock2D")


class UNet2DModel(nn.Module):
    config_name = "config.json"

    def __init__(self, **kwargs):
        super().__init__()
        cfg = dict(_DEFAULTS)
        unknown = set(kwargs) - set(cfg)
        if unknown:
            raise TypeError(f"UNet2DModel: unexpected arguments {sorted(unknown)}")
        cfg.update(kwargs)
        cfg["down_block_types"] = tuple(cfg["down_block_types"])
        cfg["up_block_types"] = tuple(cfg["up_block_types"])
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        if isinstance(cfg["sample_size"], list):
            cfg["sample_size"] = tuple(cfg["sample_size"])
        self.config = FrozenConfig(**cfg)
        c = self.config
        # the reference path uses exactly these options; anything else is refused, not approximated
        for key, want in (("time_embedding_type", "positional"), ("flip_sin_to_cos", True), ("freq_shift", 0),
                          ("act_fn", "silu"), ("resnet_time_scale_shift", "default"),
                          ("center_input_sample", False), ("class_embed_type", None), ("downsample_padding", 1),
                          ("mid_block_scale_factor", 1), ("downsample_type", "conv"), ("upsample_type", "conv")):
            if cfg[key] != want:
                raise NotImplementedError(f"UNet2DModel: {key}={cfg[key]!r} is outside the DriveSceneGen path "
                                          f"(supported: {want!r})")
        if len(c.down_block_types) != len(c.up_block_types) or len(c.down_block_types) != len(c.block_out_channels):
            raise ValueError("down_block_types, up_block_types and block_out_channels must have equal length")
        for t in c.down_block_types:
            if t not in _SUPPORTED_DOWN:
                raise NotImplementedError(f"down block type {t!r} not supported (have {_SUPPORTED_DOWN})")
        for t in c.up_block_types:
            if t not in _SUPPORTED_UP:
                raise NotImplementedError(f"up block type {t!r} not supported (have {_SUPPORTED_UP})")
        if c.norm_num_groups is None or c.attention_head_dim is None:
            raise NotImplementedError("norm_num_groups / attention_head_dim must be integers")

        boc, g, eps, lpb = c.block_out_channels, c.norm_num_groups, c.norm_eps, c.layers_per_block
        temb = boc[0] * 4
        self.conv_in = nn.Conv2d(c.in_channels, boc[0], 3, padding=1)
        self.time_embedding = _TimestepEmbedding(boc[0], temb)
        downs = []
        out_ch = boc[0]
        for i, t in enumerate(c.down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            final = i == len(boc) - 1
            res = [_Resnet(in_ch if j == 0 else out_ch, out_ch, temb, g, eps) for j in range(lpb)]
            att = [_Attention(out_ch, g, eps) for _ in range(lpb)] if t == "AttnDownBlock2D" else None
            downs.append(_Block(res, att, None if final else _Sampler(out_ch, 2), "downsamplers"))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = _Block([_Resnet(boc[-1], boc[-1], temb, g, eps) for _ in range(2)],
                                [_Attention(boc[-1], g, eps)] if c.add_attention else None, None, "")
        ups = []
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i, t in enumerate(c.up_block_types):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            final = i == len(boc) - 1
            res = []
            for j in range(lpb + 1):
                skip = in_ch if j == lpb else out_ch
                rin = prev if j == 0 else out_ch
                res.append(_Resnet(rin + skip, out_ch, temb, g, eps))
            att = [_Attention(out_ch, g, eps) for _ in range(lpb + 1)] if t == "AttnUpBlock2D" else None
            ups.append(_Block(res, att, None if final else _Sampler(out_ch, 1), "upsamplers"))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], c.out_channels, 3, padding=1)

        self.compute_dtype = "fp32"  # "fp32" | "bf16" | "fp16": see set_compute_dtype
        # dsg_unet_config.flags / DSG_UNET_BATCH_INVARIANT: row i of a batch == the batch-1 call on row i, bitwise (the
        # small-batch split-K kernels, ~20 % faster at batch 1 / 5 and equal to fp32 round-off, are then not selected)
        self.batch_invariant = False
        self._plan = None          # dsg_unet_t* (c_void_p)
        self._plan_state = {}      # name -> (data_ptr, version) last pushed
        self._plan_items = None    # cached [(state-dict key, tensor)] of this module tree
        self._plan_slots, self._plan_mods = [], []   # where each cached tensor / sub-module lives (the cache's validity check)
        self._plan_device = None
        self._ws = None

    # ---- protocol bits the reference relies on -------------------------------------------------
    @property
    def device(self):
        return self.conv_in.weight.device

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def _sample_hw(self, sample=None):
        if sample is not None:
            return int(sample.shape[-2]), int(sample.shape[-1])
        ss = self.config.sample_size
        if isinstance(ss, int):
            return ss, ss
        return int(ss[0]), int(ss[1])

    def set_compute_dtype(self, dtype):
        """Arithmetic of the convolutions / projections: "fp32" (fp32-equivalent, the default), "bf16" or "fp16" --
        what ``torch.autocast`` does to this network under accelerate's ``mixed_precision`` (train.py:24,
        training_pipeline.py:48-49): 16-bit matrix-core products and 16-bit activations between layers, fp32
        parameters / GroupNorm statistics / softmax / accumulators / outputs.  torch dtypes are accepted too."""
        names = {torch.float32: "fp32", torch.bfloat16: "bf16", torch.float16: "fp16", "no": "fp32", None: "fp32"}
        dtype = names.get(dtype, dtype)
        if dtype not in _lib.DTYPE_CODES:
            raise ValueError(f"compute dtype {dtype!r} not in {sorted(_lib.DTYPE_CODES)}")
        self.compute_dtype = dtype
        return self

    # ---- plan management -----------------------------------------------------------------------
    def _destroy_plan(self):
        if self._plan is not None:
            _lib.load().dsg_unet_destroy(self._plan)
            self._plan = None
            self._plan_state = {}

    def __del__(self):
        try:
            self._destroy_plan()
        except Exception:
            pass

    def _ensure_plan(self, h, w, device):
        lib = _lib.load()
        key = (h, w, str(device), self.compute_dtype, bool(self.batch_invariant))
        if self._plan is None or self._plan_device != key:
            self._destroy_plan()
            c = self.config
            cfg = _lib.UNetConfig()
            cfg.in_channels, cfg.out_channels = c.in_channels, c.out_channels
            cfg.sample_h, cfg.sample_w = h, w
            cfg.layers_per_block = c.layers_per_block
            cfg.num_blocks = len(c.block_out_channels)
            for i, ch in enumerate(c.block_out_channels):
                cfg.block_out_channels[i] = ch
                cfg.down_attn[i] = int(c.down_block_types[i] == "AttnDownBlock2D")
                cfg.up_attn[i] = int(c.up_block_types[i] == "AttnUpBlock2D")
            cfg.norm_num_groups = c.norm_num_groups
            cfg.norm_eps = c.norm_eps
            cfg.attention_head_dim = c.attention_head_dim
            cfg.add_attention = int(bool(c.add_attention))
            cfg.compute_dtype = _lib.DTYPE_CODES[self.compute_dtype]
            cfg.flags = _lib.UNET_BATCH_INVARIANT if self.batch_invariant else 0
            hnd = C.c_void_p()
            with torch.cuda.device(device):
                _lib.check(lib.dsg_unet_create(C.byref(cfg), C.byref(hnd)))
            self._plan = hnd
            self._plan_device = key
            self._plan_state = {}
            from .ops import sinusoid_freqs
            fr = sinusoid_freqs(c.block_out_channels[0]).to(device)
            _lib.check(lib.dsg_unet_set_param(self._plan, b"time_proj.freqs", _lib.ptr(fr), fr.numel(),
                                              _lib.stream_ptr(device)))
        # push parameters whose storage or version changed since the last push.  The (name, tensor) list is cached: building
        # `state_dict()` costs ~0.8 ms per call on the 282-tensor network -- 40 % of a batch-1 denoising step.  The cache is
        # VALIDATED on every call: each cached tensor must still be the object its module's `_parameters` /
        # `_buffers` slot holds and each cached module the object its parent's `_modules` slot holds (~40 us of dict lookups),
        # so `load_state_dict(assign=True)` through a parent, `mod.weight = nn.Parameter(..)`, `register_parameter`,
        # `torch.func.functional_call`, parametrize and a swapped sub-module all rebuild the list
        if self._plan_items is None or not self._plan_items_valid():
            self._build_plan_items()
        st = None
        for name, p in self._plan_items:
            sig = (p.data_ptr(), p._version)
            if self._plan_state.get(name) == sig:
                continue
            if p.dtype != torch.float32:
                raise RuntimeError(f"UNet2DModel: parameter {name} is {p.dtype}; the engine computes in fp32")
            if st is None:
                st = _lib.stream_ptr(device)
            d = p.detach().contiguous()
            _lib.check(lib.dsg_unet_set_param(self._plan, name.encode(), _lib.ptr(d), d.numel(), st))
            self._plan_state[name] = sig
        if st is not None:  # one host synchronisation per refresh: the weights' range-guard maxima (dsg.h)
            _lib.check(lib.dsg_unet_commit_params(self._plan))

    def _build_plan_items(self):
        self._plan_items = list(self.state_dict(keep_vars=True).items())
        want = {id(t) for _, t in self._plan_items}
        slots, mods = [], []
        for m in self.modules():
            for table in (m._parameters, m._buffers):
                for k, t in table.items():
                    if t is not None and id(t) in want:
                        slots.append((table, k, t))
            for k, child in m._modules.items():
                if child is not None:
                    mods.append((m._modules, k, child))
        self._plan_slots, self._plan_mods = slots, mods

    def _plan_items_valid(self):
        for table, k, t in self._plan_slots:
            if table.get(k) is not t:
                return False
        for table, k, m in self._plan_mods:
            if table.get(k) is not m:
                return False
        return True

    def _apply(self, fn, *a, **k):
        self._plan_items = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._plan_items = None
        return super().load_state_dict(*a, **k)

    def _workspace(self, batch, device):
        lib = _lib.load()
        need = C.c_size_t()
        _lib.check(lib.dsg_unet_workspace_bytes(self._plan, batch, C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        return self._ws

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, sample, timestep, return_dict: bool = True):
        if not sample.is_cuda:
            raise RuntimeError("drivescenegen_amd.UNet2DModel runs on the MI355X HIP engine only: move the model "
                               "and inputs to 'cuda' (there is no CPU fallback)")
        if sample.dtype != torch.float32:
            raise RuntimeError(f"UNet2DModel: input dtype {sample.dtype} not supported (fp32 engine)")
        # (both paths hand raw pointers to kernels that index by the configured channel count: a wrong shape must stop here)
        if sample.dim() != 4 or sample.shape[1] != self.config.in_channels:
            raise ValueError(f"expected a [N, {self.config.in_channels}, H, W] sample, got {tuple(sample.shape)}")
        div = 1 << (len(self.config.block_out_channels) - 1)   # (the skip connections' maps must match the up path's: diffusers fails in torch.cat)
        if sample.shape[2] % div or sample.shape[3] % div:
            raise ValueError(f"sample size {sample.shape[2]}x{sample.shape[3]} is not divisible by 2^(num_blocks-1) = {div}")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd import unet_forward_train
            out = unet_forward_train(self, sample, timestep)
        else:
            out = self._forward_plan(sample, timestep)
        if not return_dict:
            return (out,)
        return UNet2DOutput(sample=out)

    def _timesteps_tensor(self, timestep, batch, device):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=device)
        elif t.dim() == 0:
            t = t[None].to(device)
        t = t.to(device=device, dtype=torch.long)
        if t.numel() == 1 and batch != 1:
            t = t.expand(batch)
        if t.numel() != batch:
            raise ValueError(f"timesteps has {t.numel()} entries for batch {batch}")
        return t.contiguous()

    @torch.no_grad()
    def _forward_plan(self, sample, timestep):
        lib = _lib.load()
        b, c, h, w = sample.shape
        if c != self.config.in_channels:
            raise ValueError(f"expected {self.config.in_channels} input channels, got {c}")
        dev = sample.device
        with torch.cuda.device(dev):
            self._ensure_plan(h, w, dev)
            x = sample.contiguous()
            t = self._timesteps_tensor(timestep, b, dev)
            out = torch.empty((b, self.config.out_channels, h, w), dtype=torch.float32, device=dev)
            ws = self._workspace(b, dev)
            _lib.check(lib.dsg_unet_forward(self._plan, _lib.ptr(x), _lib.ptr(t), _lib.ptr(out), b,
                                            ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)))
        return out

    # ---- checkpoint I/O (training_pipeline.py:107, train.py:59) ---------------
    def save_pretrained(self, save_directory, safe_serialization: bool = False, variant=None):
        os.makedirs(save_directory, exist_ok=True)
        cfg = {"_class_name": "UNet2DModel", "_diffusers_version": "0.20.0"}
        for k, v in self.config.to_dict().items():
            cfg[k] = list(v) if isinstance(v, tuple) else v
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
            f.write("\n")
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        stem = "diffusion_pytorch_model" + (f".{variant}" if variant else "")
        if safe_serialization:
            from safetensors.torch import save_file
            save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(save_directory, stem + ".safetensors"),
                      metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(save_directory, stem + ".bin"))

    @classmethod
    def from_pretrained(cls, path, subfolder=None, variant=None, torch_dtype=None, **_unused):
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = json.load(f)
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_") and k in _DEFAULTS}
        model = cls(**cfg)
        cands = []
        if variant:  # `variant` only takes effect when such files exist (App. A.5)
            cands += [f"diffusion_pytorch_model.{variant}.safetensors", f"diffusion_pytorch_model.{variant}.bin"]
        cands += ["diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"]
        for name in cands:
            fp = os.path.join(d, name)
            if os.path.exists(fp):
                if name.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    sd = load_file(fp)
                else:
                    sd = torch.load(fp, map_location="cpu", weights_only=True)
                break
        else:
            raise FileNotFoundError(f"no diffusion_pytorch_model.[bin|safetensors] under {d}")
        legacy = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}
        fixed = {}
        for k, v in sd.items():
            for a, b in legacy.items():
                k = k.replace(a, b)
            fixed[k] = v.float()  # parameters are fp32 master copies in every mode
        model.load_state_dict(fixed, strict=True)
        if torch_dtype is not None:  # diffusers casts the module; here the dtype selects the engine's arithmetic
            model.set_compute_dtype(torch_dtype)
        return model
