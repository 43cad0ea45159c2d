#!/usr/bin/env python3
"""Build libdsg.so for gfx950 with hipcc (cross-compiles without a GPU).

    python drivescenegen_amd/csrc/build.py [--force]

Objects go to drivescenegen_amd/csrc/build/, the library to drivescenegen_amd/lib/libdsg.so
(git-ignored, but shipped to the GPU box by gpurun).
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = ["common.hip", "conv.hip", "groupnorm.hip", "attention.hip", "temb.hip", "scheduler.hip", "unet.hip", "prof.hip",
           "conv_bwd.hip", "train_ops.hip", "conv_h2.hip", "conv_h2_bf16.hip", "conv_h2_f16.hip", "imageops.hip", "raster.hip", "conv_in.hip", "conv_out.hip", "pngdec.hip", "conv_h2_gnb.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# scheduler.hip must round every fp32 operation individually (bit parity with the reference's torch-CPU
# expressions); the in-source pragma alone does not stop the backend from forming v_pk_fma_f32.
# the 16-bit conv instantiations: no SLP vectorisation -- packed fp32 VALU (v_pk_fma / v_pk_mul / v_pk_add_f32) beside MFMAs is an
# anti-lever on this part (MI355X_MICROARCH.md); measured on one box: bf16 forward 22.62 -> 22.44 ms, bf16 training step
# 58.1 -> 57.3 ms; the fp32-equivalent instantiations do not change (15.56 / 15.58 ms)
# No packed-fp32 VALU ops in the two streaming files that had them with op_sel selecting a HIGH dword for the low lane
# (v_pk_fma / mul / add_f32 ... op_sel:[0,1,..]): on gfx950 such an instruction returns wrong values on lanes 48-63 while
# another wave on the same SIMD -- of this or ANY other process -- issues v_mfma_f32_32x32x16_f16 (profiles/r04_race_under_load.txt,
# tools/probes/probe_lds_read2.hip + probe_neighbour.hip); tests/test_isa_policy.py keeps the whole library free of that form.
NO_PK_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
EXTRA_FLAGS = {"scheduler.hip": ["-ffp-contract=off"], "conv_h2_bf16.hip": ["-fno-slp-vectorize"], "conv_h2_f16.hip": ["-fno-slp-vectorize"],
               "conv_h2_gnb.hip": ["-fno-slp-vectorize"],   # (its epilogue arithmetic was packed into the op_sel hazard form with SLP on)
               "raster.hip": NO_PK_F32, "imageops.hip": NO_PK_F32}
LIB = os.path.join(ROOT, "lib", "libdsg.so")


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith(".h")] + [os.path.join(ROOT, "..", "include", "dsg.h")]
    jobs = []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(bdir, s.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = ["hipcc"] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for (src, obj), r in ex.map(cc, jobs):
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f"hipcc failed on {src}")
            if verbose:
                if r.stderr.strip():
                    sys.stderr.write(r.stderr)
                print(f"[dsg build] compiled {os.path.basename(src)}")
    objs = [os.path.join(bdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(LIB):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib", "-lz", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[dsg build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
