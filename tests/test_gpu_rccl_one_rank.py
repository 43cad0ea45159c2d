"""The data-parallel path on the hardware the GPU box has: ONE rank, but every collective issued on RCCL.
DSG_FORCE_COLLECTIVES=1 makes `Accelerator` / `GradBuckets` / bench.py create the "nccl" (= RCCL) process group under
the launcher and run the rank-0 broadcast, the bucketed asynchronous all-reduce(AVG) of the gradient slab, the barrier
and the max-over-ranks at WORLD_SIZE 1.  An average over one rank is the identity, so three training steps must
reproduce the plain single-process run BIT FOR BIT (reference: train.py:121-122 / training_pipeline.py:59-61,86 run the
same loop under accelerate's DDP).  World 2 is covered on CPU by tests/test_dist_cpu.py (gloo)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launcher(script_and_args, force, extra_env=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "DSG_FORCE_COLLECTIVES"):
        env.pop(k, None)
    if force:
        env["DSG_FORCE_COLLECTIVES"] = "1"
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
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


def test_bench_line_under_the_launcher_with_rccl_barrier_and_max():
    out = _launcher(["bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-extras"], force=True)
    rec = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert abs(rec["value"] - 16 * 3 / (rec["ms_per_step"] * 3e-3)) <= 1e-6 * rec["value"]
