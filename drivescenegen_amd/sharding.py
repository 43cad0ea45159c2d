"""Batch sharding shared by the two prepared loaders (``training._ShardedLoader`` over a torch DataLoader and
``imageops.GpuImageLoader`` over image files): accelerate's ``BatchSamplerShard(even_batches=True)`` +
``synchronize_rng_states`` semantics, which the reference gets from ``accelerator.prepare(train_dataloader)``
(/root/reference/DriveSceneGen/pipeline/training_pipeline.py:59-61; loader built at scripts/train.py:35).

* one global order per epoch, identical on every rank: rank 0 draws the epoch's seed and broadcasts it;
* rank r takes batches r, r + W, ... of that order, every batch FULL size;
* every rank runs the SAME number of steps: a ragged tail (short last batch, batch count not a multiple of W) is completed
  with samples from the start of the epoch's order -- a rank with one batch fewer would never join the last gradient
  all-reduce and the others would wait for it forever;
* ``drop_last=True``: accelerate DROPS the incomplete last round instead of completing it (``BatchSamplerShard.__len__`` =
  batches // W), so the order is cut to a multiple of batch_size * W;
* with ONE process nothing is sharded or completed: plain consecutive batches, short last batch included.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def force_collectives() -> bool:
    """DSG_FORCE_COLLECTIVES=1 under a launcher (RANK set): create the RCCL process group and issue every broadcast /
    all-reduce / barrier of the data-parallel path even when WORLD_SIZE is 1, so that one GPU exercises the calls."""
    return os.environ.get("DSG_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ


def shard_batches(order, batch_size: int, rank: int, world: int, drop_last: bool = False):
    """The index lists rank `rank` of `world` runs this epoch, given the epoch's global sample order."""
    order, b = list(order), int(batch_size)
    if drop_last:
        order = order[:len(order) // b * b]
    if world == 1:
        return [order[i:i + b] for i in range(0, len(order), b)]
    per_round = b * world
    if drop_last:  # no completion: the ragged last round goes, as in accelerate's BatchSamplerShard with drop_last
        order = order[:len(order) // per_round * per_round]
    if order and len(order) % per_round:  # even_batches: complete the last round from the start of the order
        need = per_round - len(order) % per_round
        order = order + [order[i % len(order)] for i in range(need)]
    return [order[k * b:(k + 1) * b] for k in range(rank, len(order) // b, world)]


def steps_per_epoch(n: int, batch_size: int, world: int, drop_last: bool = False) -> int:
    if drop_last:
        return n // batch_size // world
    nb = -(-n // batch_size)
    return -(-nb // world)


def broadcast_epoch_seed(rank: int, world: int, device=None) -> int:
    """Rank 0 draws a seed from the global CPU generator (what RandomSampler does) and every rank receives it."""
    if world > 1 and not dist.is_initialized():
        # (without the broadcast rank 0 would shuffle by its own seed and every other rank by 0: samples duplicated or
        # dropped across ranks with no error)
        raise RuntimeError("broadcast_epoch_seed: world > 1 but torch.distributed is not initialised -- build the "
                           "Accelerator (or call init_process_group) before iterating a sharded loader with seed=None")
    seed = torch.zeros(1, dtype=torch.int64)
    if rank == 0:
        seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
    if (world > 1 or force_collectives()) and dist.is_initialized():
        if dist.get_backend() == "nccl":
            dev_seed = seed.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
            dist.broadcast(dev_seed, src=0)
            seed = dev_seed.cpu()
        else:
            dist.broadcast(seed, src=0)
    return int(seed.item())
