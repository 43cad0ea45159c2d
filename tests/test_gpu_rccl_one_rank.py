"""The data-parallel path on the hardware the GPU box has: ONE rank, but every collective issued on RCCL.
DSG_FORCE_COLLECTIVES=1 makes `Accelerator` / `GradBuckets` / bench.py create the "nccl" (= RCCL) process group under
the launcher and run the rank-0 broadcast, the bucketed asynchronous all-reduce(AVG) of the gradient slab, the barrier
and the max-over-ranks at WORLD_SIZE 1.  An average over one rank is the identity, so three training steps must
reproduce the plain single-process run BIT FOR BIT (reference: train.py:121-122 / training_pipeline.py:59-61,86 run the
same loop under accelerate's DDP).  World 2 runs on the same GPU through gloo (two processes, one device: RCCL refuses that,
gloo carries the CUDA tensors through the host) -- there the all-reduce changes the gradients, and the host logic of world 2
is covered on CPU by tests/test_dist_cpu.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launcher(script_and_args, force, extra_env=None, nproc=1):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "DSG_FORCE_COLLECTIVES"):
        env.pop(k, None)
    if force:
        env["DSG_FORCE_COLLECTIVES"] = "1"
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(port)] + script_and_args
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def _lines(out, key):
    return [ln for ln in out.splitlines() if ln.startswith(key)]


def test_training_steps_over_one_rank_rccl_are_bitwise_the_plain_run():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DSG_FORCE_COLLECTIVES")}
    plain = subprocess.run([sys.executable, os.path.join("tools", "ddp_smoke.py")], env=env, cwd=ROOT,
                           capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-4000:]
    forced = _launcher([os.path.join("tools", "ddp_smoke.py")], force=True)
    assert _lines(plain.stdout, "collectives")[0].split()[1] == "off"
    f = _lines(forced, "collectives")[0].split()
    assert f[1] == "on" and f[3] == "nccl" and int(f[5]) >= 1, f   # the buckets exist and went through RCCL
    assert len(_lines(forced, "rank 0 of 1")) == 3
    assert _lines(forced, "rank") == _lines(plain.stdout, "rank")     # the three losses, as hex floats
    assert _lines(forced, "checksum") == _lines(plain.stdout, "checksum")


def test_default_net_buckets_launch_back_to_front_inside_the_backward_walk():
    """The train.py:39-57 network (56.6 M parameters, 226 MB gradient slab = 8 buckets of >= 25 MB) through the same path
    (VERDICT r02 item 4): the buckets reach RCCL in strictly descending order (the backward walk finalises the last layers'
    gradients first), all but the last-filled ones are launched from INSIDE the walk -- a GPU event recorded at the first
    launch precedes the end-of-walk event by a measurable stretch of backward work -- and two training steps are bitwise
    the plain single-process run (an all-reduce(AVG) over one rank is the identity; stream-ordering bugs between the
    backward stream and RCCL's communication stream would show as different bits or a hang)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DSG_FORCE_COLLECTIVES")}
    args = [os.path.join("tools", "ddp_smoke.py"), "DEFAULT3", "2", "2"]
    plain = subprocess.run([sys.executable] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-4000:]
    forced = _launcher(args, force=True, extra_env={"DSG_DDP_TRACE": "1"})
    f = _lines(forced, "collectives")[0].split()
    assert f[1] == "on" and f[3] == "nccl" and int(f[5]) >= 8, f
    tr = json.loads(_lines(forced, "trace")[0][len("trace "):])
    order = tr["order"]
    assert len(order) == int(f[5]) and order == sorted(order, reverse=True) and len(set(order)) == len(order), order
    assert sum(tr["in_walk"]) >= len(order) - 1, tr["in_walk"]          # (bucket 0 holds conv_in: final with the walk's last kernel)
    assert tr["in_walk"][0] and tr["ms_before_walk_end"][0] > 1.0, tr   # the first all-reduce has milliseconds of backward to hide under
    assert _lines(forced, "rank") == _lines(plain.stdout, "rank")
    assert _lines(forced, "checksum") == _lines(plain.stdout, "checksum")


def _floats(lines, key="loss"):
    return [float.fromhex(ln.split()[ln.split().index(key) + 1]) for ln in lines]


def test_two_ranks_on_the_one_gpu_average_their_gradients():
    """World size 2 on the hardware the box has: two processes share the GPU, the collectives go through gloo (RCCL refuses two
    ranks on one device; DSG_DIST_BACKEND is a test hook).  Everything else is the product path: the rank-0 parameter
    broadcast, the bucketed all-reduce launched asynchronously from inside the backward walk of the HIP tape, SUM + the
    1 / world scale kernel, clip, AdamW.  Unlike the one-rank run, the all-reduce CHANGES the gradient slab here, so a missing
    stream dependency between the backward walk, the collective and the optimizer would show as wrong bits.
      (a) both ranks on the same batch: (g + g) / 2 = g exactly -- losses and final parameters bitwise the one-process run's,
          on both ranks;
      (b) the batch split between the ranks (train.py:121-122 under accelerate: each process its own shard): the mean of the
          two ranks' step-0 losses is the whole-batch loss, the parameters after the update agree between the ranks bit for
          bit, and the next step's mean loss follows the one-process run (the update differs by the rounding of a sum)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DSG_FORCE_COLLECTIVES")}
    args = [os.path.join("tools", "ddp_smoke.py"), "CFG1", "4", "3"]
    plain = subprocess.run([sys.executable] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-4000:]
    p_loss = _floats(_lines(plain.stdout, "rank"))
    # (a)
    same = _launcher(args, force=False, extra_env={"DSG_DIST_BACKEND": "gloo"}, nproc=2)
    f = _lines(same, "collectives")
    assert len(f) == 2 and all(x.split()[1] == "on" and x.split()[3] == "gloo" and int(x.split()[5]) >= 1 for x in f), f
    for r in (0, 1):
        mine = [ln.replace(f"rank {r} of 2", "rank 0 of 1") for ln in _lines(same, f"rank {r} of 2")]
        assert mine == _lines(plain.stdout, "rank"), (r, mine)
    assert _lines(same, "checksum") == _lines(plain.stdout, "checksum") * 2
    # (b)
    split = _launcher(args, force=False, extra_env={"DSG_DIST_BACKEND": "gloo", "DSG_SMOKE_SHARD": "1"}, nproc=2)
    l0, l1 = _floats(_lines(split, "rank 0 of 2")), _floats(_lines(split, "rank 1 of 2"))
    assert len(l0) == 3 and len(l1) == 3 and l0 != l1                      # different shards
    cs = _lines(split, "checksum")
    assert len(cs) == 2 and cs[0] == cs[1]                                 # one model on both ranks, bit for bit
    assert abs((l0[0] + l1[0]) / 2 - p_loss[0]) <= 2e-6 * p_loss[0]        # same parameters, the batch in two halves
    for i in (1, 2):   # after one / two averaged updates
        assert abs((l0[i] + l1[i]) / 2 - p_loss[i]) <= 2e-3 * p_loss[i], (i, l0, l1, p_loss)


def test_two_ranks_default_net_eight_buckets_overlap_the_backward_walk():
    """The train.py:39-57 network at world size 2 on the one GPU (gloo, as above): 8 buckets of >= 25 MB leave in descending
    order from inside the backward walk on both ranks, the averaged step is bitwise the one-process step (both ranks on the
    same batch), and the two ranks hold the same parameters afterwards."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DSG_FORCE_COLLECTIVES")}
    args = [os.path.join("tools", "ddp_smoke.py"), "DEFAULT3", "2", "2"]
    plain = subprocess.run([sys.executable] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-4000:]
    two = _launcher(args, force=False, extra_env={"DSG_DIST_BACKEND": "gloo", "DSG_DDP_TRACE": "1"}, nproc=2)
    f = _lines(two, "collectives")
    assert len(f) == 2 and all(x.split()[3] == "gloo" and int(x.split()[5]) >= 8 for x in f), f
    traces = [json.loads(ln[len("trace "):]) for ln in _lines(two, "trace")]
    assert len(traces) == 2
    for tr in traces:
        order = tr["order"]
        assert order == sorted(order, reverse=True) and len(set(order)) == len(order) >= 8, order
        assert sum(tr["in_walk"]) >= len(order) - 1 and tr["in_walk"][0], tr["in_walk"]
    for r in (0, 1):
        mine = [ln.replace(f"rank {r} of 2", "rank 0 of 1") for ln in _lines(two, f"rank {r} of 2")]
        assert mine == _lines(plain.stdout, "rank"), (r, mine)
    assert _lines(two, "checksum") == _lines(plain.stdout, "checksum") * 2


def test_overlapped_and_deferred_buckets_give_the_same_step_bitwise():
    """DSG_DDP_OVERLAP / Accelerator.ddp_overlap: the same buckets all-reduced from inside the backward walk and after it.
    tools/ddp_smoke.py's self-check runs step 0 both ways and compares loss and every updated parameter bit for bit -- on the
    one-rank RCCL communicator (56.6 M parameters, 8 buckets, the stream hand-over between the walk and RCCL's stream) and at
    world size 2 on the shared GPU (gloo), where the all-reduce changes the slab.  With DSG_DDP_OVERLAP=0 in the environment a
    whole run launches nothing from inside the walk, and its steps are bitwise the default run's."""
    args = [os.path.join("tools", "ddp_smoke.py"), "DEFAULT3", "2", "2", "--selfcheck"]
    one = _launcher(args, force=True)
    sc = _lines(one, "selfcheck")
    assert len(sc) == 1 and sc[0].endswith("OK") and "loss_equal True params_equal True" in sc[0] and "buckets 8" in sc[0], sc
    two = _launcher([os.path.join("tools", "ddp_smoke.py"), "CFG1", "4", "2"], force=False,
                    extra_env={"DSG_DIST_BACKEND": "gloo", "DSG_SMOKE_SHARD": "1"}, nproc=2)   # (world > 1: unprompted)
    sc = _lines(two, "selfcheck")
    assert len(sc) == 2 and all(x.endswith("OK") and "params_equal True" in x for x in sc), sc
    # the environment switch on a whole run: no bucket leaves from inside the walk, same bits
    on = _launcher(args[:-1], force=True, extra_env={"DSG_DDP_TRACE": "1", "DSG_DDP_SELFCHECK": "0"})
    off = _launcher(args[:-1], force=True, extra_env={"DSG_DDP_TRACE": "1", "DSG_DDP_OVERLAP": "0", "DSG_DDP_SELFCHECK": "0"})
    tr_on, tr_off = (json.loads(_lines(o, "trace")[0][len("trace "):]) for o in (on, off))
    assert sum(tr_on["in_walk"]) >= len(tr_on["order"]) - 1 and not any(tr_off["in_walk"]), (tr_on, tr_off)
    assert tr_on["order"] == tr_off["order"]
    assert _lines(on, "rank") == _lines(off, "rank") and _lines(on, "checksum") == _lines(off, "checksum")


def test_bench_line_under_the_launcher_with_rccl_barrier_and_max():
    out = _launcher(["bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-extras"], force=True)
    rec = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert abs(rec["value"] - 16 * 3 / (rec["ms_per_step"] * 3e-3)) <= 1e-6 * rec["value"]


def test_bench_line_of_two_ranks_sharing_the_gpu():
    """`bench.py --gpus 2` as the driver launches it, with both ranks on the box's one GPU (gloo, the test hook): the N > 1 code of
    the contract really runs -- barrier on both sides of the timed region, each rank's own wall time gathered, the MAXIMUM taken,
    rank 0 alone printing ONE line whose value is the samples of BOTH ranks over that time."""
    out = _launcher(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-extras"], force=False,
                    extra_env={"DSG_DIST_BACKEND": "gloo"}, nproc=2)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, len(lines)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["rccl_world"] == 2
    per = rec["per_rank_ms_per_step"]
    assert len(per) == 2 and abs(rec["ms_per_step"] - max(per)) <= 1e-6 * max(per)
    assert abs(rec["value"] - 2 * 16 * 3 / (rec["ms_per_step"] * 3e-3)) <= 1e-6 * rec["value"]
    assert rec["config"]["global_batch"] == 32


def _ddp_record(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, len(lines)
    assert len(lines[0]) < 6000
    rec = json.loads(lines[0])
    return rec, rec["summary"]


def test_train_ddp_record_on_a_one_rank_rccl_group():
    """whenever a process group exists `bench.py` also emits a `train_ddp` record -- the data-parallel
    TRAINING step through Accelerator + GradBuckets on RCCL, timed with the buckets off / overlapped / deferred, with the
    bitwise self-check of the two modes -- so the first multi-GPU run measures the exchange north_star names.  Here: the
    forced one-rank RCCL communicator (an average over one rank is the identity; every collective is still issued)."""
    out = _launcher(["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras", "--no-prof",
                     "--ddp-fp32-batch", "4", "--ddp-bf16-batch", "4", "--ddp-steps", "2"], force=True)
    rec, summ = _ddp_record(out)
    for name in ("train_ddp_fp32", "train_ddp_bf16"):
        r = summ[name]
        assert "error" not in r, r
        assert r["world"] == 1 and r["backend"] == "nccl" and r["buckets"] == 8 and r["selfcheck_bitwise"] is True
        assert r["value"] > 0 and set(r["ms"]) == {"local", "overlap", "deferred"} and all(v > 0 for v in r["ms"].values())
        assert abs(r["value"] - 4 * 1 / (r["ms"]["overlap"] * 1e-3)) <= 1e-6 * r["value"]
        assert set(r["exposed_comm_ms"]) == {"overlap", "deferred"}


def test_train_ddp_record_of_two_ranks_sharing_the_gpu():
    """World 2 on the box's one GPU over gloo: the all-reduce really mixes two different shards' gradients (rank-specific
    samples), both modes still give the same slab bit for bit on both ranks, the global rate counts both ranks."""
    out = _launcher(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras", "--no-prof",
                     "--ddp-fp32-batch", "2", "--ddp-bf16-batch", "0", "--ddp-steps", "2"], force=False,
                    extra_env={"DSG_DIST_BACKEND": "gloo"}, nproc=2)
    rec, summ = _ddp_record(out)
    r = summ["train_ddp_fp32"]
    assert "error" not in r, r
    assert r["world"] == 2 and r["backend"] == "gloo" and r["buckets"] == 8 and r["selfcheck_bitwise"] is True
    assert abs(r["value"] - 2 * 2 / (r["ms"]["overlap"] * 1e-3)) <= 1e-6 * r["value"]
    assert "train_ddp_bf16" not in summ and rec["n_gpus"] == 2
