"""Training-side counterparts of what DriveSceneGen's train loop takes from torch / accelerate.

Reference: /root/reference/DriveSceneGen/pipeline/training_pipeline.py:46-107 and
/root/reference/DriveSceneGen/scripts/train.py:66-71:
    loss = F.mse_loss(noise_pred, noise)                       -> mse_loss
    accelerator.backward(loss)                                 -> Accelerator.backward
    accelerator.clip_grad_norm_(model.parameters(), 1.0)       -> clip_grad_norm_
    optimizer = torch.optim.AdamW(model.parameters(), lr)      -> AdamW (fused, flat slabs)
    Accelerator(mixed_precision, gradient_accumulation_steps, log_with, project_dir) + .prepare/.accumulate/
    .log/.init_trackers/.unwrap_model/.is_main_process          -> Accelerator
    (latent) DistributedDataParallel gradient averaging         -> GradBuckets over RCCL

All arithmetic (loss, norms, clipping, optimizer update) runs in libdsg.so; this file is host logic.
"""
from __future__ import annotations

import json
import math
import os
import time
from contextlib import contextmanager

import torch
import torch.distributed as dist

from . import ops, sharding
from .autograd import SLABS, get_train_state


# ------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------
class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        loss, dpred = ops.mse_loss(pred.detach(), target.detach())
        ctx.dpred = dpred
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        # gout is 1 for a plain backward; loss scaling / accumulation factors stay on the device
        d = ops.scale(ctx.dpred, gout.reshape(1).contiguous(), 1.0, out=ctx.dpred)
        return d, None


def mse_loss(pred, target):
    """``F.mse_loss(pred, target)`` (mean reduction, training_pipeline.py:85) on the HIP engine."""
    if not pred.is_cuda:
        raise RuntimeError("drivescenegen_amd.mse_loss runs on the MI355X HIP engine only (got a CPU tensor)")
    return _MSEFn.apply(pred, target)


# ------------------------------------------------------------------------------------------------
# gradient clipping + optimizer over flat slabs
# ------------------------------------------------------------------------------------------------
def _slab_of(params):
    """If every parameter's .grad is a slice of ONE contiguous fp32 buffer (the UNet's TrainState slab),
    return (flat_grad, [(param, offset, numel)])."""
    params = [p for p in params if p.grad is not None]
    if not params:
        return None, []
    base = params[0].grad.untyped_storage().data_ptr()
    flat = SLABS.get(base)
    if flat is None:
        return None, []
    flat = flat[0]
    rows = []
    for p in params:
        g = p.grad
        if g.untyped_storage().data_ptr() != base or not g.is_contiguous() or g.dtype != torch.float32:
            return None, []
        rows.append((p, (g.data_ptr() - base) // 4, g.numel()))
    return flat, rows


def clip_grad_norm_(parameters, max_norm: float, norm_type: float = 2.0):
    """``torch.nn.utils.clip_grad_norm_`` semantics (L2, eps 1e-6): returns the total norm (0-d device tensor)
    and scales the gradients in place by min(1, max_norm / (norm + 1e-6))."""
    if norm_type != 2.0:
        raise NotImplementedError("clip_grad_norm_: only the L2 norm (the reference's default) is implemented")
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return torch.zeros(())
    if not params[0].grad.is_cuda:
        raise RuntimeError("drivescenegen_amd.clip_grad_norm_ runs on the MI355X HIP engine only")
    flat, rows = _slab_of(params)
    # the one-kernel path is taken only when the passed parameters are EVERY slice of the slab: for a proper subset
    # (frozen layers, param groups) the norm must not see -- and the scaling must not touch -- the other gradients
    if flat is not None and len(rows) == SLABS[flat.untyped_storage().data_ptr()][1]:
        total = ops.l2_norm(flat)  # slab padding is zero
        ops.clip_scale_(flat, total, max_norm)
    else:  # gradients living in separate tensors: one norm per tensor, combined on the device
        sq = torch.stack([ops.l2_norm(p.grad.contiguous().view(-1)) for p in params]).view(-1).contiguous()
        total = ops.l2_norm(sq)
        for p in params:
            ops.clip_scale_(p.grad.view(-1), total, max_norm)
    return total.view(())


class AdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW`` (decoupled weight decay; torch defaults betas (0.9, 0.999), eps 1e-8,
    weight_decay 1e-2 -- train.py:66 passes only lr) with ONE fused kernel per step over flat
    parameter / gradient / moment slabs when the parameters belong to a drivescenegen_amd UNet2DModel."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("AdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = {}  # id(group) -> dict(param_flat, m, v, grad_flat, step)

    def _flatten(self, group):
        params = [p for p in group["params"] if p.grad is not None]
        gflat, rows = _slab_of(params)
        if gflat is None or len(params) != len(group["params"]):
            return None
        pflat = torch.zeros_like(gflat)
        for p, off, n in rows:  # move the parameters into one slab laid out like the gradient slab
            pflat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = pflat[off:off + n].view(p.shape)
        return dict(param=pflat, grad=gflat, m=torch.zeros_like(gflat), v=torch.zeros_like(gflat), step=0)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            key = id(group)
            st = self._flat.get(key)
            if st is None:
                st = self._flatten(group) or False
                self._flat[key] = st
            if st:
                st["step"] += 1
                ops.adamw_step_(st["param"], st["grad"], st["m"], st["v"], st["step"], group["lr"], group["betas"],
                                group["eps"], group["weight_decay"])
                self._bump_versions(group["params"])
                continue
            for p in group["params"]:  # separate tensors: one launch per parameter
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("drivescenegen_amd.AdamW runs on the MI355X HIP engine only")
                s = self.state[p]
                if not s:
                    s["step"], s["exp_avg"], s["exp_avg_sq"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                s["step"] += 1
                ops.adamw_step_(p.data.view(-1), p.grad.contiguous().view(-1), s["exp_avg"].view(-1),
                                s["exp_avg_sq"].view(-1), s["step"], group["lr"], group["betas"], group["eps"],
                                group["weight_decay"])
            self._bump_versions(group["params"])
        return loss

    @staticmethod
    def _bump_versions(params):
        # the update happened behind autograd's back (raw pointers): bump version counters so that cached
        # engine-layout weights (UNet2DModel._plan_state / TrainState.versions) are refreshed
        for p in params:
            torch.autograd.graph.increment_version(p)

    def zero_grad(self, set_to_none: bool = False):
        """Keeps .grad attached to the flat slab (one memset) instead of dropping the tensors."""
        done = set()
        for group in self.param_groups:
            flat, rows = _slab_of([p for p in group["params"] if p.grad is not None])
            if flat is not None and flat.data_ptr() not in done:
                flat.zero_()
                done.add(flat.data_ptr())
            elif flat is None:
                for p in group["params"]:
                    if p.grad is not None:
                        p.grad.zero_()


# ------------------------------------------------------------------------------------------------
# data-parallel gradient averaging: reverse-order flat buckets, launched as they fill
# ------------------------------------------------------------------------------------------------
_force_collectives = sharding.force_collectives


class GradBuckets:
    """All-reduce(mean) of a flat gradient slab in ~bucket_mb buckets.  ``ready(name)`` is called by the
    backward pass when a parameter's gradient is final; a bucket is launched asynchronously on the
    communication stream the moment its last gradient lands (buckets fill back-to-front, so communication
    overlaps the rest of backward).  ``finish()`` waits for the outstanding work.

    ``overlap=False`` (environment: DSG_DDP_OVERLAP=0) is the A/B switch for a first multi-GPU run: the same buckets, in
    the same order, are launched from ``finish()`` -- after the backward walk -- so nothing of RCCL runs beside the
    backward kernels.  The reduced values cannot depend on WHEN a bucket is launched; if the two modes ever differ on real
    xGMI links, the overlap (RCCL's reduction kernels beside the 16-deep-MFMA backward kernels) is what to look at
    (tools/ddp_smoke.py --selfcheck compares them bitwise)."""

    def __init__(self, flat, offsets: dict, group=None, bucket_mb: float = 25.0, overlap=None):
        self.flat, self.group = flat, group
        self.overlap = (os.environ.get("DSG_DDP_OVERLAP", "1") != "0") if overlap is None else bool(overlap)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a one-rank RCCL communicator still runs every call below (DSG_FORCE_COLLECTIVES=1: the one-GPU hardware test)
        self.active = self.world > 1 or (dist.is_initialized() and _force_collectives())
        names = sorted(offsets, key=lambda k: offsets[k][0])
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets, self.of = [], {}
        lo, cnt, members = None, 0, []
        for nm in names:
            off, n = offsets[nm]
            if lo is None:
                lo = off
            members.append(nm)
            cnt = off + n - lo
            if cnt >= cap:
                self.buckets.append(dict(lo=lo, hi=off + n, names=members))
                lo, members = None, []
        if members:
            self.buckets.append(dict(lo=lo, hi=offsets[members[-1]][0] + offsets[members[-1]][1], names=members))
        for i, b in enumerate(self.buckets):
            for nm in b["names"]:
                self.of[nm] = i
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.avg_native = backend == "nccl"  # RCCL has ReduceOp.AVG; gloo does not
        self.last_launch_order = []
        self.reset()

    def reset(self):
        self.pending = [len(b["names"]) for b in self.buckets]
        self.seen = set()
        self.works = []
        self.launch_order = []
        self.deferred = []    # overlap off: buckets complete during the walk, launched by finish() in completion order
        self._in_finish = False

    # DSG_DDP_TRACE=1 (tests / tools): keep, per step, which buckets were launched from inside the backward walk and a
    # GPU event at each launch and at the end of the walk -- "communication overlaps backward" as a measured statement
    trace_enabled = os.environ.get("DSG_DDP_TRACE") == "1"
    last_trace = None

    def ready(self, name):
        if not self.active or name in self.seen or name not in self.of:
            return
        self.seen.add(name)
        i = self.of[name]
        self.pending[i] -= 1
        if self.pending[i] == 0:
            if self.overlap:
                self._launch(i)
            else:
                self.deferred.append(i)

    def _launch(self, i):
        b = self.buckets[i]
        view = self.flat[b["lo"]:b["hi"]]
        op = dist.ReduceOp.AVG if self.avg_native else dist.ReduceOp.SUM
        ev = None
        if self.trace_enabled and view.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()   # on the backward walk's stream: everything enqueued so far precedes the bucket's all-reduce
        self.works.append((dist.all_reduce(view, op=op, group=self.group, async_op=True), view))
        self.launch_order.append(i)
        if self.trace_enabled:
            self._trace = getattr(self, "_trace", [])
            self._trace.append((i, not self._in_finish, ev))

    def finish(self):
        if not self.active:
            return
        self._in_finish = True
        end_ev = None
        if self.trace_enabled and self.flat.is_cuda:
            end_ev = torch.cuda.Event(enable_timing=True)
            end_ev.record()   # the end of the backward walk on its stream
        for i in self.deferred:                  # overlap off: the walk's completion order, launched now
            self._launch(i)
        self.deferred = []
        for i, left in enumerate(self.pending):  # parameters that got no gradient this step
            if left > 0:
                self.pending[i] = 0
                self._launch(i)
        for work, view in self.works:
            work.wait()
            if not self.avg_native:
                if view.is_cuda:
                    ops.scale(view, None, 1.0 / self.world, out=view)
                else:  # gloo on CPU tensors: only reachable from the host-logic tests
                    view.div_(self.world)
        if self.trace_enabled:
            tr = getattr(self, "_trace", [])
            if end_ev is not None:
                torch.cuda.synchronize()
            GradBuckets.last_trace = dict(
                order=[i for i, _, _ in tr], in_walk=[w for _, w, _ in tr],
                ms_before_walk_end=[(ev.elapsed_time(end_ev) if ev is not None and end_ev is not None else None)
                                    for _, _, ev in tr])
            self._trace = []
        self.last_launch_order = list(self.launch_order)   # (kept across reset(): what the step just finished did)
        self.reset()


# ------------------------------------------------------------------------------------------------
# accelerate.Accelerator, the subset training_pipeline.py:48-61,82-101 uses
# ------------------------------------------------------------------------------------------------
class _ShardedLoader:
    """accelerate's prepared DataLoader (``BatchSamplerShard`` with ``even_batches=True`` + ``synchronize_rng_states``):
    every rank gets its own batches of the FULL batch size -- rank r takes batches r, r+W, ... of one global batch
    order that is identical on all ranks -- and every rank runs the SAME number of steps per epoch: when the batch
    count is not a multiple of the world size (or the last batch is short) the tail is completed with samples from
    the start of the epoch's order, so no rank waits in a collective that the others never join.
    With ONE process accelerate shards nothing: the loader runs as it is -- its own sampler (generator, replacement,
    num_samples) and a SHORT last batch, exactly the reference's single-GPU epoch (train.py:35: batch 14, shuffle=True).

    The global order comes from the loader's own sampler when it is sequential; for a shuffling loader
    (train.py:35 ``shuffle=True``) rank 0 draws a seed from the global CPU generator and broadcasts it, and every
    rank builds the same permutation from it -- the shards are then a partition of the dataset."""

    def __init__(self, loader, device, rank, world):
        self.loader, self.device, self.rank, self.world = loader, device, rank, world
        self.dataset = getattr(loader, "dataset", None)
        self.batch_size = getattr(loader, "batch_size", None)
        self.drop_last = bool(getattr(loader, "drop_last", False))
        sampler = getattr(loader, "sampler", None)
        self.shuffle = isinstance(sampler, torch.utils.data.RandomSampler)
        self.index_mode = (self.dataset is not None and self.batch_size is not None and hasattr(self.dataset, "__getitem__")
                           and isinstance(sampler, (torch.utils.data.RandomSampler, torch.utils.data.SequentialSampler)))
        self.epoch_seed = None  # the seed the current epoch's permutation was built from (tests / resume)

    def _num_batches(self):
        n, b = len(self.dataset), self.batch_size
        return n // b if self.drop_last else math.ceil(n / b)

    def __len__(self):
        if self.world == 1:
            return len(self.loader)
        if self.index_mode:
            return sharding.steps_per_epoch(len(self.dataset), self.batch_size, self.world, self.drop_last)
        return math.ceil(len(self.loader) / self.world)

    def _to_device(self, batch):
        return batch.to(self.device, non_blocking=True) if torch.is_tensor(batch) else batch

    def _global_order(self):
        n = len(self.dataset)
        if not self.shuffle:
            return list(range(n))
        self.epoch_seed = sharding.broadcast_epoch_seed(self.rank, self.world, self.device)
        g = torch.Generator()
        g.manual_seed(self.epoch_seed)
        return torch.randperm(n, generator=g).tolist()

    def __iter__(self):
        if self.world == 1:  # nothing to shard, nothing to even out: the loader's own batches, short tail included
            for batch in self.loader:
                yield self._to_device(batch)
            return
        if not self.index_mode:  # opaque loader: rank r keeps every W-th batch, the tail wraps to the first batches
            head, count = [], 0
            for i, batch in enumerate(self.loader):
                if i < self.world:
                    head.append(batch)
                count += 1
                if i % self.world == self.rank:
                    yield self._to_device(batch)
            for j in range(count, math.ceil(count / self.world) * self.world):
                if j % self.world == self.rank:
                    yield self._to_device(head[(j - count) % len(head)])
            return
        collate = self.loader.collate_fn
        for idx in sharding.shard_batches(self._global_order(), self.batch_size, self.rank, self.world, self.drop_last):
            yield self._to_device(collate([self.dataset[i] for i in idx]))


class _SteppedScheduler:
    """accelerate's AcceleratedScheduler: one call advances the wrapped scheduler `world` times (the schedule
    was sized on the unsharded loader, train.py:67-71) and is skipped when the optimizer step was skipped."""

    def __init__(self, sched, world, accel):
        self.sched, self.world, self.accel = sched, world, accel

    def step(self, *a, **k):
        if not self.accel.sync_gradients or self.accel.optimizer_step_was_skipped:
            return
        for _ in range(self.world):
            self.sched.step(*a, **k)

    def get_last_lr(self):
        return self.sched.get_last_lr()

    def __getattr__(self, n):
        return getattr(self.sched, n)


class GradScaler:
    """``torch.cuda.amp.GradScaler`` as accelerate drives it under ``mixed_precision='fp16'`` (train.py:24; SURVEY
    App. A.6): the loss is multiplied by ``scale`` before backward, gradients are un-scaled (and checked for
    inf / nan) before clipping, a step whose gradients are not finite is SKIPPED and halves the scale, 2000 clean
    steps in a row double it.  The arithmetic (scaling, the finite check) runs in libdsg.so."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.scale = float(init_scale)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self._growth_tracker = 0
        self._unscaled = False
        self._found = None  # device int32 flag of the current step

    def get_scale(self):
        return self.scale

    def unscale_(self, params):
        if self._unscaled:
            return
        params = [p for p in params if p.grad is not None]
        if not params:
            return
        dev = params[0].grad.device
        self._found = torch.zeros(1, dtype=torch.int32, device=dev)
        flat, rows = _slab_of(params)
        if flat is not None and len(rows) == SLABS[flat.untyped_storage().data_ptr()][1]:
            ops.unscale_check_(flat, 1.0 / self.scale, self._found)
        else:
            for p in params:
                ops.unscale_check_(p.grad.view(-1), 1.0 / self.scale, self._found)
        self._unscaled = True

    def step(self, optimizer, params):
        """Runs ``optimizer.step()`` unless this step's gradients hold an inf / nan; returns True when skipped."""
        self.unscale_(params)
        skipped = self._found is not None and bool(self._found.item())  # (one D2H sync per step, as torch's scaler)
        if not skipped:
            optimizer.step()
        return skipped

    def update(self, skipped):
        if skipped:
            self.scale *= self.backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self.growth_interval:
                self.scale *= self.growth_factor
                self._growth_tracker = 0
        self._unscaled, self._found = False, None

    def state_dict(self):
        return {"scale": self.scale, "growth_tracker": self._growth_tracker}

    def load_state_dict(self, sd):
        self.scale, self._growth_tracker = float(sd["scale"]), int(sd["growth_tracker"])


class _PreparedOptimizer:
    """accelerate's ``AcceleratedOptimizer``: ``step`` / ``zero_grad`` do nothing on micro-batches whose gradients
    are still being accumulated (``accelerator.sync_gradients`` False), and under fp16 the step goes through the
    GradScaler (skipped when the gradients are not finite -- the LR scheduler then skips too)."""

    def __init__(self, optimizer, accel):
        self.optimizer, self.accel = optimizer, accel

    def _params(self):
        return [p for g in self.optimizer.param_groups for p in g["params"]]

    def step(self, closure=None):
        a = self.accel
        if not a.sync_gradients:
            return None
        if a.scaler is not None:
            skipped = a.scaler.step(self.optimizer, self._params())
            a.scaler.update(skipped)
            a.optimizer_step_was_skipped = skipped
            return None
        a.optimizer_step_was_skipped = False
        return self.optimizer.step(closure) if closure is not None else self.optimizer.step()

    def zero_grad(self, set_to_none=None):
        if self.accel.sync_gradients:
            self.optimizer.zero_grad()

    @property
    def step_was_skipped(self):
        return self.accel.optimizer_step_was_skipped

    def __getattr__(self, n):
        return getattr(self.optimizer, n)


_MIXED = {"no": "fp32", None: "fp32", "fp16": "fp16", "bf16": "bf16"}


class Accelerator:
    def __init__(self, mixed_precision="no", gradient_accumulation_steps=1, log_with=None, project_dir=None,
                 device=None):
        # mixed_precision (train.py:24 ships 'fp16'; BASELINE configs[4] names bf16): the prepared model runs its
        # convolutions / projections on the f16 / bf16 matrix cores with 16-bit activations in HBM and fp32
        # GroupNorm statistics, softmax, accumulators, master weights and optimizer -- torch.autocast's split.
        # 'fp16' adds the GradScaler (skipped steps on inf / nan); 'bf16' needs none; 'no' is the fp32-equivalent engine.
        if mixed_precision not in _MIXED:
            raise ValueError(f"mixed_precision={mixed_precision!r}")
        self.mixed_precision = mixed_precision or "no"
        self.scaler = GradScaler() if self.mixed_precision == "fp16" else None
        self.optimizer_step_was_skipped = False
        self.gradient_accumulation_steps = int(gradient_accumulation_steps)
        self.project_dir = project_dir
        self.log_with = log_with
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.collectives = self.world > 1 or _force_collectives()
        if self.collectives and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(self._device_index())
            # RCCL ("nccl" on ROCm) is the data path.  DSG_DIST_BACKEND=gloo is a test hook: gloo carries CUDA tensors through
            # the host, which lets TWO ranks share ONE GPU (RCCL refuses that) -- the only way to run the world-2 step on a
            # one-GPU box (tests/test_gpu_rccl_one_rank.py)
            dist.init_process_group(os.environ.get("DSG_DIST_BACKEND", "nccl"))
        self.device = torch.device(device) if device else torch.device("cuda", self._device_index())
        self.sync_gradients = True
        self._accum = 0
        self._buckets = None
        self.ddp_overlap = None   # None: DSG_DDP_OVERLAP (default on); True / False: this Accelerator's choice (A/B switch)
        self._model = None
        self._log_file = None
        self.trackers = []

    def _device_index(self):
        # one rank per GPU; more local ranks than GPUs (the gloo test hook above) wrap around
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        return self.local_rank % n if n else self.local_rank

    @property
    def is_main_process(self):
        return self.rank == 0

    @property
    def is_local_main_process(self):
        return self.local_rank == 0

    @property
    def num_processes(self):
        return self.world

    @property
    def process_index(self):        # (accelerate's names for rank / local rank)
        return self.rank

    @property
    def local_process_index(self):
        return self.local_rank

    def init_trackers(self, project_name, config=None):
        if self.project_dir and self.is_main_process:
            os.makedirs(self.project_dir, exist_ok=True)
            self._log_file = open(os.path.join(self.project_dir, f"{project_name}.jsonl"), "a")

    def log(self, values, step=None):
        if self._log_file:
            rec = {"step": step, "time": time.time()}
            rec.update({k: (float(v) if hasattr(v, "__float__") else v) for k, v in values.items()})
            self._log_file.write(json.dumps(rec) + "\n")
            self._log_file.flush()

    def end_training(self):
        if self._log_file:
            self._log_file.close()
            self._log_file = None

    def wait_for_everyone(self):
        if self.collectives:
            dist.barrier()

    def unwrap_model(self, model):
        return model

    def prepare(self, *objs):
        out = []
        for o in objs:
            if isinstance(o, torch.nn.Module):
                o.to(self.device)
                self._model = o
                if hasattr(o, "set_compute_dtype"):
                    o.set_compute_dtype(_MIXED[self.mixed_precision])
                if self.collectives:
                    for p in o.parameters():  # identical replicas: rank 0's initial weights
                        dist.broadcast(p.data, src=0)
                out.append(o)
            elif isinstance(o, torch.utils.data.DataLoader):
                out.append(_ShardedLoader(o, self.device, self.rank, self.world))
            elif isinstance(o, torch.optim.Optimizer):
                out.append(_PreparedOptimizer(o, self))
            elif isinstance(o, torch.optim.lr_scheduler.LRScheduler):
                out.append(_SteppedScheduler(o, self.world, self))
            else:
                out.append(o)
        return tuple(out) if len(out) > 1 else out[0]

    @contextmanager
    def accumulate(self, model=None):
        self._accum += 1
        self.sync_gradients = self._accum % self.gradient_accumulation_steps == 0
        yield

    def _ensure_buckets(self):
        if not self.collectives or self._model is None:
            return None
        st = get_train_state(self._model)
        if self._buckets is None or self._buckets.flat.data_ptr() != st.grad_flat.data_ptr():
            self._buckets = GradBuckets(st.grad_flat, st.offsets, overlap=self.ddp_overlap)
            st.grad_ready_hooks[:] = [self._on_ready]
            st.post_backward[:] = [self._finish_buckets]  # (runs when the backward walk is done, before its loss scale is removed)
        elif self.ddp_overlap is not None:
            self._buckets.overlap = bool(self.ddp_overlap)
        return self._buckets

    def _finish_buckets(self):
        if self.sync_gradients and self._buckets is not None:
            self._buckets.finish()

    def _on_ready(self, name):
        if self.sync_gradients and self._buckets is not None:
            self._buckets.ready(name)

    def backward(self, loss):
        self._ensure_buckets()
        mult = 1.0 / self.gradient_accumulation_steps
        if self.scaler is not None:
            mult *= self.scaler.get_scale()
        if mult != 1.0:
            loss = _ScaleLoss.apply(loss, mult)
        loss.backward()  # (the tape's end-of-backward callback finishes the bucketed all-reduce: _finish_buckets)

    def clip_grad_norm_(self, parameters, max_norm, norm_type=2):
        if not self.sync_gradients:
            return None
        parameters = list(parameters)
        if self.scaler is not None:  # accelerate: scaler.unscale_(optimizer) before the clip
            self.scaler.unscale_(parameters)
        return clip_grad_norm_(parameters, max_norm, float(norm_type))


class _ScaleLoss(torch.autograd.Function):
    """loss / gradient_accumulation_steps without a torch arithmetic kernel on the path."""

    @staticmethod
    def forward(ctx, loss, mult):
        ctx.mult = mult
        return ops.scale(loss.detach().reshape(1).contiguous(), None, mult).view(())

    @staticmethod
    def backward(ctx, g):
        return ops.scale(g.reshape(1).contiguous(), None, ctx.mult).view(()), None
