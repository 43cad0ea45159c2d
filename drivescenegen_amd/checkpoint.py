"""Full training resume.

The reference saves only the pipeline folder every epoch (training_pipeline.py:106-107) and its "resume" is a
commented-out ``UNet2DModel.from_pretrained(output_dir, subfolder="unet")`` (train.py:59): optimizer moments, LR
schedule position, epoch / step and RNG state are lost.  ``save_training_state`` / ``load_training_state`` add one
file next to the unchanged diffusers layout: ``<output_dir>/training_state.pt``.
"""
from __future__ import annotations

import os

import torch

STATE_FILE = "training_state.pt"


def _optimizer_state(optimizer):
    flat = [s for s in getattr(optimizer, "_flat", {}).values() if s]
    if flat:  # drivescenegen_amd.AdamW on flat slabs: one entry per param group
        return {"kind": "flat", "groups": [dict(m=s["m"].cpu(), v=s["v"].cpu(), step=s["step"]) for s in flat],
                "hyper": [{k: v for k, v in g.items() if k != "params"} for g in optimizer.param_groups]}
    return {"kind": "torch", "state": optimizer.state_dict()}


def save_training_state(output_dir, optimizer, lr_scheduler=None, epoch=0, global_step=0, extra=None):
    os.makedirs(output_dir, exist_ok=True)
    sched = getattr(lr_scheduler, "sched", lr_scheduler)
    state = {"optimizer": _optimizer_state(optimizer),
             "lr_scheduler": sched.state_dict() if sched is not None else None,
             "epoch": int(epoch), "global_step": int(global_step),
             "cpu_rng": torch.get_rng_state(), "extra": extra or {}}
    tmp = os.path.join(output_dir, STATE_FILE + ".tmp")
    torch.save(state, tmp)
    os.replace(tmp, os.path.join(output_dir, STATE_FILE))


def load_training_state(output_dir, optimizer, lr_scheduler=None, restore_rng=True):
    """Restores optimizer / scheduler state in place; returns (epoch, global_step, extra).  For the flat-slab AdamW
    the parameters must already have their gradient slab (i.e. one backward has run, or call
    ``drivescenegen_amd.autograd.get_train_state(model)`` first)."""
    state = torch.load(os.path.join(output_dir, STATE_FILE), map_location="cpu", weights_only=False)
    opt = state["optimizer"]
    if opt["kind"] == "flat":
        for group, g_saved, hyper in zip(optimizer.param_groups, opt["groups"], opt["hyper"]):
            st = optimizer._flat.get(id(group))
            if not st:
                st = optimizer._flatten(group)
                if not st:
                    raise RuntimeError("load_training_state: parameters have no flat gradient slab yet")
                optimizer._flat[id(group)] = st
            st["m"].copy_(g_saved["m"])
            st["v"].copy_(g_saved["v"])
            st["step"] = int(g_saved["step"])
            group.update({k: v for k, v in hyper.items() if k != "initial_lr" or "initial_lr" not in group})
    else:
        optimizer.load_state_dict(opt["state"])
    sched = getattr(lr_scheduler, "sched", lr_scheduler)
    if sched is not None and state["lr_scheduler"] is not None:
        sched.load_state_dict(state["lr_scheduler"])
    if restore_rng:
        torch.set_rng_state(state["cpu_rng"])
    return state["epoch"], state["global_step"], state["extra"]
