"""The C-ABI library builds for gfx950, loads without a GPU, and exports every symbol include/dsg.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "dsg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dsg_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(lib_built):
    lib = ctypes.CDLL(lib_built)
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dsg.h but not exported by libdsg.so"


def test_python_binding_table_matches_header(lib_built):
    from drivescenegen_amd import _lib
    declared = set(_header_symbols())
    bound = set(_lib.SIGNATURES) | set(_lib.OTHER_SYMBOLS)
    assert declared <= bound, declared - bound
    assert bound - declared <= {"dsg_conv2d_fwd_direct"}  # test hook, not part of the public header


def test_error_path_without_gpu(lib_built):
    """Argument validation happens before any HIP call, so it is checkable on a CPU-only host."""
    from drivescenegen_amd import _lib
    lib = _lib.load()
    assert lib.dsg_version() >= 100
    rc = lib.dsg_conv2d_fwd(None, None)
    assert rc == -1 and b"NULL" in lib.dsg_last_error()
    rc = lib.dsg_attention_fwd(1, 1, 1, 7, 2, 16, None)  # 7 channels / 2 heads
    assert rc == -1 and b"divisible" in lib.dsg_last_error()
    rc = lib.dsg_ddim_step(1, 1, 1, 0, 0.0, 0.0, 0.0, 0.0, 0.0, None)
    assert rc == -1


def test_cpu_tensor_is_refused_loudly(lib_built):
    import pytest
    import torch
    import drivescenegen_amd as d
    m = d.UNet2DModel(sample_size=32, block_out_channels=(32, 64), down_block_types=("DownBlock2D",) * 2,
                      up_block_types=("UpBlock2D",) * 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 32, 32), 0)
    with pytest.raises(RuntimeError, match="HIP engine"):
        d.DDPMScheduler().add_noise(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8), torch.tensor([1]))


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under drivescenegen_amd/ (or the C sources) may import, call or
    even name it, and bench.py only does so inside its cpu_baseline leg."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "drivescenegen_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), (dirpath, f)
    bench = open(os.path.join(root, "bench.py")).read()
    hits = [m.start() for m in re.finditer(r"^\s*from oracle\.", bench, re.M)]
    leg = bench.index("def cpu_leg(")
    end = bench.index("\ndef ", leg + 1)
    assert hits and all(leg < h < end for h in hits)


def test_bench_line_helpers_quote_stamped_counter_files_and_digest_every_record():
    """bench.py's host-side pieces that need no GPU: the committed PMC files carry the git head and the command they were
    collected with and every kernel the roofline objects name is found in them (a renamed instantiation would read None);
    `summary_of` -- the line's last key -- keeps value / ms_per_step / roofline.frac of every record, failed ones included."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dsg_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for kern in ("dsg::conv_h2_kernel<0, 4, 3, 2, 4, 1, 3, 64, 0, 0, 0, 0>", "dsg::conv_h2_kernel<0, 2, 3, 2, 4, 2, 3, 64, 0, 1, 0, 0>",
                 "dsg::conv_h2_kernel<0, 4, 3, 2, 4, 1, 3, 64, 0, 0, 1, 0>", "dsg::conv_h2_kernel<0, 2, 3, 2, 4, 2, 3, 64, 0, 1, 1, 0>"):
        t = bench.pmc_traffic(kern)
        assert t["traffic"] and t["traffic_git_head"] not in (None, "unknown") and "bench.py" in t["traffic_cmd"], (kern, t)
    assert bench.pmc_mfma("dsg::conv_h2_kernel<0, 4, 3, 2, 4, 1, 3, 64, 0, 0, 0, 0>")["mfma_pipe_util"] > 0.3
    for pat, sfx in ((r"conv_wgrad_h2w?_kernel", "_train_fp32"), (r"conv_wgrad16_kernel<\d, 3", "_train_bf16"),
                     (r"conv_h2_kernel<0, 2, 3, 2, 4, 2, 3, 64, 0, 1, 0", "_cfg4"), (r"conv_h2_kernel<0, 4, 3, [02], 4, 1, 3, (64|128), 1, 0, 0", "_bf16")):
        t = bench.pmc_class_traffic(pat, sfx)
        assert t["traffic"] and t["traffic_git_head"] not in (None, "unknown"), (pat, t)
    out = {"value": 1000.0, "unit": "image-steps/s", "ms_per_step": 16.0, "roofline": {"bound": "mfma", "frac": 0.5, "second_kernel": {"frac": 0.4}},
           "step_ms_spread": {"min": 15.9, "median": 16.0, "max": 16.2, "n": 20},
           "extra_records": {"train_fp32": {"value": 250.0, "ms_per_step": 256.0, "config": {"batch": 64}, "roofline": {"frac": 0.38}},
                             "broken": {"error": "RuntimeError: " + "x" * 500}},
           "cpu_baseline": {"value": 2.0, "unit": "image-steps/s", "cores": 16, "kind": "port", "sample": "..."}}
    s = bench.summary_of(out)
    assert s["headline"]["value"] == 1000.0 and s["headline"]["roofline"]["second_kernel_frac"] == 0.4
    assert s["headline"]["step_ms_min_med_max"] == [15.9, 16.0, 16.2]
    assert s["train_fp32"]["batch"] == 64 and s["train_fp32"]["roofline"]["frac"] == 0.38
    assert len(s["broken"]["error"]) <= 200 and s["cpu_baseline"]["kind"] == "port"
