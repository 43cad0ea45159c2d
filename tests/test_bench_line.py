"""bench.py's stdout line is what the driver parses: it must stay small, carry the contract's keys, and quote counter
traffic that is possible (>= the algorithmic bytes of the same class).  BENCH_r04.json had "parsed": null because the
line had grown to 25.6 KB."""
import importlib.util
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _canned():
    """a full record bench.py produced on the GPU box (round 4's: 25 KB, six extra records)"""
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))


def test_compact_line_is_small_and_has_the_contract_keys():
    b = _bench()
    full = _canned()
    assert len(json.dumps(full)) > 20000          # the thing the driver could not parse
    full["summary"] = b.summary_of(full)
    text = b.compact_line(full, "gpurun_out/bench_full.json")
    assert "\n" not in text and len(text) < 6000, len(text)
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_git_head"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-12
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert "workload" in line["config"] and "model" not in line["config"]
    # one digest entry per extra record, each with its number
    for name in full["extra_records"]:
        assert name in line["summary"], name
    assert line["summary"]["train_fp32"]["ms_per_step"] == full["extra_records"]["train_fp32"]["ms_per_step"]


def test_compact_line_survives_failed_legs_and_long_text():
    b = _bench()
    full = _canned()
    full["extra_records"] = {k: {"error": "RuntimeError: " + "x" * 5000} for k in full["extra_records"]}
    full["dtype"] = "f32 " * 500
    full["config"]["workload"] = "w" * 3000
    full["cpu_baseline"]["sample"] = "s" * 3000
    full["summary"] = b.summary_of(full)
    text = b.compact_line(full, None)
    assert len(text) < 6000
    line = json.loads(text)
    assert line["value"] == full["value"] and line["summary"]["train_bf16"]["error"].startswith("RuntimeError")


def test_counter_traffic_is_at_least_the_algorithmic_bytes():
    """Every kernel class bench.py prices with a PMC pattern: the launch-weighted HBM bytes of the instantiations the
    pattern selects must be >= the class's algorithmic bytes per launch (r04's mixed_bf16 record quoted 438.8 MB for a
    class of 652.7 MB: the pattern missed the two-per-CU instantiations of the same HIP-event class)."""
    b = _bench()
    full = _canned()
    checked = 0
    for rec_name, pats, suffix in (("mixed_bf16", b.MIXED_PMC_PATTERNS, "_bf16"), ("configs3_512", b.CFG4_PMC_PATTERNS, "_cfg4")):
        rows = full["extra_records"][rec_name]["kernels"]
        for cls, pat in pats.items():
            if cls not in rows:
                continue
            t = b.pmc_class_traffic(pat, suffix)
            if t.get("traffic") is None:
                continue
            assert t["traffic"] >= 0.98 * rows[cls]["bytes_per_launch"], (rec_name, cls, t["traffic"], rows[cls]["bytes_per_launch"])
            checked += 1
    assert checked >= 4
    # and the patterns select what the HIP-event class holds: same launch count per step in counters and events
    d = json.load(open(sorted(p for p in (os.path.join(ROOT, "profiles", f) for f in os.listdir(os.path.join(ROOT, "profiles")))
                              if p.endswith("_pmc_traffic_bf16.json"))[-1]))
    n = sum(v["launches"] for k, v in d["kernels"].items() if re.search(b.MIXED_PMC_PATTERNS["conv3x3_s1_mfma_16bit"], k))
    assert n > 0


def test_upsample_class_carries_its_footnote():
    b = _bench()
    assert "4/9" in b.CLASS_NOTES["conv3x3_upsample_mfma_f16x2split"]


def test_round6_record_digests_the_end_to_end_loop_and_the_ddp_step():
    """The records round 6 added must reach the line the driver parses: `summary.train_e2e` (files -> loader -> train_steps at
    the two training configs' own batch, next to the bare tape: VERDICT r05 item 1) from a full record of the GPU box, and
    `summary.train_ddp_*` (the data-parallel step with the buckets off / overlapped / deferred: item 3) -- still under 6 KB."""
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    full["train_ddp"] = {"fp32": {"value": 1990.0, "unit": "images/s", "batch_per_gpu": 64, "world": 8, "backend": "nccl",
                                  "ms": {"local": 254.0, "overlap": 257.0, "deferred": 262.0},
                                  "exposed_comm_ms": {"overlap": 3.0, "deferred": 8.0}, "buckets": 8, "selfcheck_bitwise": True,
                                  "per_rank_ms": {"overlap": [257.0] * 8}, "config": {"workload": "w" * 300}},
                         "bf16": {"error": "RuntimeError: " + "x" * 2000}}
    full["summary"] = b.summary_of(full)
    text = b.compact_line(full, "gpurun_out/bench_full.json")
    assert len(text) < 6000, len(text)
    line = json.loads(text)
    e2e = line["summary"]["train_e2e"]
    assert set(e2e) == {"fp32_b64", "bf16_b128"}
    for key, rec in e2e.items():
        assert 0.95 <= rec["device_noise_vs_bare"] <= 1.05, (key, rec)        # the verdict's bar for the device-noise mode
        assert 0.9 <= rec["host_noise_vs_bare"] <= 1.05 and rec["loader_images_s"] > 5 * rec["pil_one_thread_images_s"]
        assert abs(rec["device_noise_images_s"] / rec["bare_tape_images_s"] - rec["device_noise_vs_bare"]) < 2e-3
    assert "BARE TAPE" in full["extra_records"]["train_bf16"]["metric"]
    ddp = line["summary"]["train_ddp_fp32"]
    assert ddp["world"] == 8 and ddp["exposed_comm_ms"] == {"overlap": 3.0, "deferred": 8.0} and ddp["selfcheck_bitwise"] is True
    assert "per_rank_ms" not in ddp and line["summary"]["train_ddp_bf16"]["error"].startswith("RuntimeError")
