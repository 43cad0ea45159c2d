"""N>1 host logic on CPU with gloo, world_size 2: bucketed gradient averaging, dataloader
sharding, LR-scheduler stepping, and collective-free sample sharding of the sampler."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from drivescenegen_amd.training import GradBuckets, _ShardedLoader, _SteppedScheduler
        # --- bucketed all-reduce(mean) over a flat slab, launched back-to-front as gradients land ---
        offsets, off = {}, 0
        for i, n in enumerate([1000, 64, 5000, 3, 70000, 128, 9000]):
            offsets[f"p{i}"] = (off, n)
            off += (n + 63) // 64 * 64
        flat = torch.zeros(off)
        for nm, (o, n) in offsets.items():
            flat[o:o + n] = (rank + 1) * (1 + int(nm[1:]))
        b = GradBuckets(flat, offsets, bucket_mb=0.01)
        assert len(b.buckets) >= 3
        for nm in reversed(list(offsets)):  # backward order
            b.ready(nm)
        order = list(b.launch_order)
        b.finish()
        for nm, (o, n) in offsets.items():
            assert torch.allclose(flat[o:o + n], torch.full((n,), 1.5 * (1 + int(nm[1:])))), nm
        assert order == sorted(order, reverse=True), order  # last bucket first: overlap with backward
        # overlap off (DSG_DDP_OVERLAP=0): nothing leaves before finish(); same buckets, same order, same averages
        for nm, (o, n) in offsets.items():
            flat[o:o + n] = (rank + 1) * (1 + int(nm[1:]))
        b2 = GradBuckets(flat, offsets, bucket_mb=0.01, overlap=False)
        for nm in reversed(list(offsets)):
            b2.ready(nm)
        assert b2.launch_order == [] and b2.deferred == order
        b2.finish()
        assert b2.last_launch_order == order
        for nm, (o, n) in offsets.items():
            assert torch.allclose(flat[o:o + n], torch.full((n,), 1.5 * (1 + int(nm[1:])))), nm
        # a step where one parameter gets no gradient still completes
        flat.fill_(float(rank))
        for nm in list(offsets)[1:]:
            b.ready(nm)
        b.finish()
        for nm, (o, n) in offsets.items():  # (slab padding between buckets carries no gradient and is not reduced)
            assert torch.allclose(flat[o:o + n], torch.full((n,), 0.5)), nm
        # --- dataloader sharding (accelerate's even_batches + synchronised shuffle): every rank runs the same number
        #     of full-size batches; the shards partition the dataset, the ragged tail wraps to the epoch's start ---
        from drivescenegen_amd.imageops import GpuImageLoader
        ds = torch.arange(40, dtype=torch.float32).view(20, 2)   # 20 samples, batch 4 -> 5 batches: odd for world 2
        for shuffle in (False, True):
            dl = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=shuffle)
            sl = _ShardedLoader(dl, "cpu", rank, world)
            for _epoch in range(2):
                mine = [x for x in sl]
                assert len(mine) == len(sl) == 3 and all(x.shape == (4, 2) for x in mine)
                got = torch.cat(mine)[:, 0] / 2                   # sample ids of this rank, in order
                allv = [torch.zeros(12) for _ in range(world)]
                dist.all_gather(allv, got)
                flat = torch.stack(allv)                          # [rank][3 batches x 4]
                order = torch.stack([flat[r].view(3, 4) for r in range(world)], 1).reshape(-1)  # batch k of rank r = global batch k*W + r
                assert torch.equal(order[:20].sort().values, torch.arange(20, dtype=torch.float32))  # a partition
                assert torch.equal(order[20:], order[:4])          # the tail is the start of the same order
                if shuffle:
                    seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
                    dist.all_gather(seeds, torch.tensor([sl.epoch_seed]))
                    assert seeds[0].item() == seeds[1].item()
                else:
                    assert torch.equal(order[:20], torch.arange(20, dtype=torch.float32))
        # drop_last: 22 samples, batch 4 -> 5 full batches; accelerate's BatchSamplerShard(drop_last=True) DROPS the
        # incomplete last round (len = 5 // 2): two steps per rank, samples 0..15 seen once, nothing completed
        dl = torch.utils.data.DataLoader(torch.arange(22, dtype=torch.float32).view(22, 1), batch_size=4, drop_last=True)
        sl = _ShardedLoader(dl, "cpu", rank, world)
        mine = torch.cat([x for x in sl])[:, 0]
        assert len(sl) == 2 and mine.numel() == 8
        allv = [torch.zeros(8) for _ in range(world)]
        dist.all_gather(allv, mine)
        assert torch.equal(torch.cat(allv).sort().values, torch.arange(16, dtype=torch.float32))
        gl = GpuImageLoader([f"f{i}.png" for i in range(22)], (8, 8), batch_size=4, shuffle=False, device="cpu", rank=rank,
                            world=world, drop_last=True)
        assert len(gl) == 2 and [len(b) for b in gl._batches()] == [4, 4]
        # --- the GPU input pipeline's loader shards the same way (host logic only here: file indices, no decode):
        #     7 batches of 3 from 21 files -> odd for world 2, plus seed=None -> rank 0's broadcast seed ---
        for seed in (None, 11):
            gl = GpuImageLoader([f"f{i}.png" for i in range(20)], (8, 8), batch_size=3, shuffle=True, seed=seed, device="cpu",
                                rank=rank, world=world)
            for _epoch in range(2):
                mine = gl._batches()
                gl.epoch += 1
                assert len(mine) == len(gl) == 4 and all(len(b) == 3 for b in mine)
                allv = [torch.zeros(12, dtype=torch.int64) for _ in range(world)]
                dist.all_gather(allv, torch.tensor(mine).reshape(-1))
                order = torch.stack([v.view(4, 3) for v in allv], 1).reshape(-1)   # global batch k*W + r
                assert torch.equal(order[:20].sort().values, torch.arange(20))      # the shards partition the files
                assert torch.equal(order[20:], order[:4])                           # the tail wraps to the epoch's start
                seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
                dist.all_gather(seeds, torch.tensor([gl.epoch_seed]))
                assert seeds[0].item() == seeds[1].item()
        # --- gradient accumulation: the prepared optimizer steps / zeroes only on the synchronising micro-batch ---
        from drivescenegen_amd.training import Accelerator, _PreparedOptimizer

        class CountingOpt:
            param_groups = []
            steps = zeros = 0

            def step(self):
                self.steps += 1

            def zero_grad(self):
                self.zeros += 1
        acc = Accelerator.__new__(Accelerator)
        acc.gradient_accumulation_steps, acc._accum, acc.sync_gradients = 3, 0, True
        acc.scaler, acc.optimizer_step_was_skipped = None, False
        copt = CountingOpt()
        popt = _PreparedOptimizer(copt, acc)
        sched_steps = []
        lam = _SteppedScheduler(type("S", (), {"step": lambda self: sched_steps.append(1)})(), world, acc)
        for micro in range(6):
            with acc.accumulate():
                popt.step()
                lam.step()
                popt.zero_grad()
        assert copt.steps == 2 and copt.zeros == 2 and len(sched_steps) == 2 * world
        # --- accelerate semantics: one lr_scheduler.step() advances the schedule `world` times ---
        from drivescenegen_amd.optimization import get_cosine_schedule_with_warmup
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        sch = get_cosine_schedule_with_warmup(opt, 10, 100)

        class A:
            sync_gradients = True
            optimizer_step_was_skipped = False
        st = _SteppedScheduler(sch, world, A())
        opt.step()
        st.step()
        assert abs(st.get_last_lr()[0] - world / 10) < 1e-12
        # --- sampler sharding: rank r keeps rows [r*B/W, (r+1)*B/W) of the full-batch CPU noise stream ---
        from drivescenegen_amd.pipelines import _PipelineBase
        full = torch.randn(6, 3, 4, 4, generator=torch.manual_seed(14555))
        part = _PipelineBase._shard(full, (rank, world))
        parts = [torch.zeros_like(part) for _ in range(world)]
        dist.all_gather(parts, part)
        assert torch.equal(torch.cat(parts), full)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + repr(e) + "\n" + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_world2_gloo(lib_built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_dataset_png_and_pkl(tmp_path):
    """Image_Dataset: PNG and .pkl branches, bilinear no-antialias resize, [-1,1] normalisation
    (reference dataset.py:21-24,37-45)."""
    import numpy as np
    from PIL import Image
    from types import SimpleNamespace
    from drivescenegen_amd.dataset import Image_Dataset
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    Image.fromarray(a).save(tmp_path / "a.png")
    torch.save({"fig_tensor": torch.from_numpy(a).float() / 255}, tmp_path / "b.pkl")
    cfg = SimpleNamespace(dataset_name=str(tmp_path / "*"), patterns_size_height=16, patterns_size_width=16)
    ds = Image_Dataset(cfg)
    assert len(ds) == 2
    want = torch.nn.functional.interpolate((torch.from_numpy(a).permute(2, 0, 1).float() / 255)[None], size=(16, 16),
                                           mode="bilinear", align_corners=False)[0]
    want = (want - 0.5) / 0.5
    for i in range(2):
        x = ds[i]
        assert x.shape == (3, 16, 16) and x.dtype == torch.float32
        assert torch.allclose(x, want, atol=1e-6)
        assert float(x.min()) >= -1 and float(x.max()) <= 1


def test_grad_scaler_state_machine():
    """GradScaler bookkeeping (host logic; the unscale / finite check kernel is covered by the gpu tests): a skipped
    step halves the scale and resets the streak, `growth_interval` clean steps double it (torch defaults 65536 /
    2.0 / 0.5 / 2000)."""
    from drivescenegen_amd.training import GradScaler
    sc = GradScaler(growth_interval=3)
    assert sc.get_scale() == 65536.0
    sc.update(skipped=True)
    assert sc.get_scale() == 32768.0
    for _ in range(2):
        sc.update(skipped=False)
    assert sc.get_scale() == 32768.0
    sc.update(skipped=False)
    assert sc.get_scale() == 65536.0
    sc.update(skipped=False)
    sc.update(skipped=True)   # a skip in the middle of a streak restarts it
    for _ in range(2):
        sc.update(skipped=False)
    assert sc.get_scale() == 32768.0
    sd = sc.state_dict()
    other = GradScaler()
    other.load_state_dict(sd)
    assert other.get_scale() == sc.get_scale() and other._growth_tracker == sc._growth_tracker


def test_one_process_loader_is_the_loader_itself():
    """accelerate shards nothing when num_processes == 1: the prepared loader yields the wrapped loader's own batches --
    short last batch included, the sampler's own generator honoured -- as the reference's single-GPU run does
    (train.py:35: DataLoader(dataset, batch_size=14, shuffle=True))."""
    from drivescenegen_amd.training import _ShardedLoader
    ds = torch.arange(23, dtype=torch.float32).view(23, 1)

    def make():
        return torch.utils.data.DataLoader(ds, batch_size=5, shuffle=True, generator=torch.Generator().manual_seed(7))
    want = [b.clone() for b in make()]
    sl = _ShardedLoader(make(), "cpu", 0, 1)
    got = list(sl)
    assert len(sl) == len(got) == 5 and [tuple(b.shape) for b in got] == [(5, 1)] * 4 + [(3, 1)]
    assert all(torch.equal(a, b) for a, b in zip(got, want))          # the sampler's generator decides the order
    assert torch.equal(torch.cat(got).flatten().sort().values, ds.flatten())   # every sample exactly once: no padded tail
    # sampling with replacement / num_samples passes through untouched too
    smp = torch.utils.data.RandomSampler(ds, replacement=True, num_samples=12, generator=torch.Generator().manual_seed(3))
    sl = _ShardedLoader(torch.utils.data.DataLoader(ds, batch_size=5, sampler=smp), "cpu", 0, 1)
    assert [b.shape[0] for b in sl] == [5, 5, 2]


def test_sharded_loader_without_a_process_group_raises_and_bad_pickles_do_not_recurse(tmp_path):
    """(1) a sharded loader with seed=None iterated before the process group exists must not let every rank
    shuffle by its own seed; (2) a directory whose .pkl files hold no dict ends in a clear error, not a RecursionError;
    (3) a one-process drop_last epoch keeps the loader's own count."""
    import pytest
    from drivescenegen_amd import sharding
    from drivescenegen_amd.imageops import GpuImageLoader
    with pytest.raises(RuntimeError, match="not initialised"):
        sharding.broadcast_epoch_seed(1, 2)
    assert sharding.steps_per_epoch(22, 4, 1, True) == 5 and sharding.steps_per_epoch(22, 4, 2, True) == 2
    assert sharding.steps_per_epoch(22, 4, 2, False) == 3 and sharding.steps_per_epoch(22, 4, 4, True) == 1
    assert [len(sharding.shard_batches(range(22), 4, r, 4, True)) for r in range(4)] == [1, 1, 1, 1]
    for i in range(3):
        torch.save([i], tmp_path / f"{i}.pkl")
    ld = GpuImageLoader(str(tmp_path / "*.pkl"), (8, 8), batch_size=1, device="cpu")
    with pytest.raises(IndexError, match="no usable sample"):
        ld._load_one(0)
    torch.save({"fig_tensor": torch.ones(4, 4, 3)}, tmp_path / "9.pkl")
    ld = GpuImageLoader(str(tmp_path / "*.pkl"), (8, 8), batch_size=1, device="cpu")
    assert ld._load_one(0).shape == (4, 4, 3)     # the three non-dict pickles are skipped


def test_noise_drawn_ahead_is_the_serial_loops_noise():
    """train_loop.NoiseAhead / batches_with_noise (reference training_pipeline.py:72): the noise of step
    k+1 is drawn on a worker thread while step k runs, and the stream of values -- and of every other consumer of the global
    CPU generator: the shuffling sampler's per-epoch seed, a dataset that draws in __getitem__ -- is the serial loop's."""
    from drivescenegen_amd.train_loop import batches_with_noise

    class Jitter(torch.utils.data.Dataset):     # draws from the global generator when a sample is fetched
        def __len__(self):
            return 11

        def __getitem__(self, i):
            return torch.full((2, 4, 4), float(i)) + torch.rand(())

    def run(overlap):
        torch.manual_seed(77)
        dl = torch.utils.data.DataLoader(Jitter(), batch_size=3, shuffle=True)   # 4 batches, the last one short
        out = []
        for _epoch in range(2):
            if overlap is None:     # the reference's loop
                for batch in dl:
                    out.append((batch, torch.randn(batch.shape)))
            else:
                out.extend(batches_with_noise(dl, overlap))
        return out
    ref = run(None)
    for overlap in (True, False):
        got = run(overlap)
        assert len(got) == len(ref) == 8
        for (b0, n0), (b1, n1) in zip(ref, got):
            assert torch.equal(b0, b1) and torch.equal(n0, n1) and n1.shape == b1.shape


def test_noise_drawn_ahead_is_undone_when_the_consumer_stops_early():
    """`batches_with_noise` has step k+1's draw in flight when it yields step k.  A consumer that stops there
    (an exception or `break` in its loop, a max-steps cut) must leave the global CPU generator where the SERIAL loop would
    have left it -- the epoch-end DDPM sample and the next epoch's sampler seed draw from it next."""
    from drivescenegen_amd.train_loop import batches_with_noise
    batches = [torch.zeros(2, 3, 8, 8) for _ in range(5)]

    def run(overlap, stop, how):
        torch.manual_seed(123)
        seen = []
        try:
            for i, (_b, n) in enumerate(batches_with_noise(batches, overlap)):
                seen.append(n.clone())
                if i == stop:
                    if how == "raise":
                        raise KeyError("consumer failed")
                    break
        except KeyError:
            pass
        return seen, torch.randn(4)          # the generator's next consumer

    for how in ("break", "raise"):
        for stop in (0, 2, 4):
            s_seen, s_next = run(False, stop, how)
            o_seen, o_next = run(True, stop, how)
            assert len(o_seen) == stop + 1 and all(torch.equal(a, b) for a, b in zip(s_seen, o_seen))
            assert torch.equal(s_next, o_next), (how, stop)


def test_untrusted_pickles_that_need_full_unpickling_are_skipped_like_non_dicts(tmp_path):
    """with the restricted unpickler a .pkl holding a non-allowlisted object raises instead of returning a
    non-dict -- the reference skips such a file for the next one (utils/datasets/dataset.py:37-39).  Both loaders treat the
    refusal as 'not a usable dict' (bounded by one round of the list), and `Image_Dataset` no longer unpickles arbitrary
    objects from the data directory unless the config says trust_pickles=True."""
    import pickle
    from types import SimpleNamespace
    from drivescenegen_amd.dataset import Image_Dataset
    from drivescenegen_amd.imageops import GpuImageLoader

    with open(tmp_path / "0.pkl", "wb") as f:     # an object that is not on torch's weights_only allowlist
        pickle.dump(SimpleNamespace(fig_tensor=[[1.0]]), f)
    torch.save([1, 2, 3], tmp_path / "1.pkl")                                    # a non-dict the restricted unpickler accepts
    torch.save({"fig_tensor": torch.full((4, 4, 3), 0.25)}, tmp_path / "2.pkl")
    files = str(tmp_path / "*.pkl")
    ld = GpuImageLoader(files, (8, 8), batch_size=1, device="cpu")
    ld.files = sorted(ld.files)
    assert float(ld._load_one(0).mean()) == 0.25                                 # files 0 and 1 skipped, file 2 used
    cfg = SimpleNamespace(dataset_name=files, patterns_size_height=8, patterns_size_width=8)
    ds = Image_Dataset(cfg)
    ds.data_list = sorted(ds.data_list)
    x = ds[0]
    assert x.shape == (3, 8, 8) and torch.allclose(x, torch.full((3, 8, 8), -0.5))   # (0.25 - 0.5) / 0.5
    os.remove(tmp_path / "2.pkl")
    ds = Image_Dataset(cfg)
    with pytest.raises(IndexError, match="no usable sample"):
        ds[0]


def test_device_noise_streams_are_rank_disjoint_and_resumable():
    """train_loop.DeviceNoise (the opt-in replacement of training_pipeline.py:72's host draw): step k on rank r draws the tensor
    named (seed, offset = r << 40 | k) -- different ranks never share an offset, a run resumed at step k continues the same
    sequence, and the oracle's streams for two such names really differ (host logic + the numpy restatement: no GPU)."""
    import numpy as np
    from drivescenegen_amd.train_loop import DeviceNoise
    from oracle import philox_oracle as po
    a, b = DeviceNoise(seed=3, rank=0), DeviceNoise(seed=3, rank=1)
    offs_a, offs_b = [a.next_offset() for _ in range(6)], [b.next_offset() for _ in range(6)]
    assert offs_a == list(range(6)) and offs_b == [(1 << 40) | k for k in range(6)]
    assert not set(offs_a) & set(offs_b)
    resumed = DeviceNoise(seed=3, rank=1, step=4)
    assert [resumed.next_offset(), resumed.next_offset()] == offs_b[4:6]
    s0, s1 = po.stream_u32(64, 3, offs_a[1]), po.stream_u32(64, 3, offs_b[1])
    assert not np.array_equal(s0, s1) and np.array_equal(s0, po.stream_u32(64, 3, 1))
