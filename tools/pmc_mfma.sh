#!/bin/bash
# Matrix-core utilisation of the bench's kernels from the SQ counters (one timeout 900 rocprofv3 --pmc pass, kernel-trace only).
# Writes gpurun_out/<tag>_pmc_mfma.json: per kernel, launches and per-launch means of SQ_INSTS_MFMA (MFMA instructions
# issued, summed over waves), SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE (GPU-clock cycles the kernel
# was resident), and two utilisation figures:
#   issued  = SQ_INSTS_MFMA x 32 cycles (a 32x32x16 f16 MFMA holds its SIMD's matrix pipe for 8 passes x 4 cycles)
#             / (cycles x 1024 SIMDs)                    -- independent of the clock the chip throttles to
#   counter = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024) -- the gfx94x MfmaUtil formula (ROCm 7.2 ships no gfx950
#             derived-counter section; the guide says the gfx94x ones are what applies)
# with cycles = GRBM_GUI_ACTIVE / 8: the counter is reported summed over the 8 XCDs (checked: GUI_ACTIVE / 8 / kernel
# duration = 1.99 GHz, the clock the chip runs at under this load; SQ_INSTS_MFMA equals 3 x algorithmic FLOPs / 32768
# to the last digit, and SQ_VALU_MFMA_BUSY_CYCLES = 32 x SQ_INSTS_MFMA).
# Usage (on the GPU box): bash tools/pmc_mfma.sh <tag> [suffix]     (PMC_CMD / suffix as in pmc_bench.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1
export PMC_SUFFIX=${2:-}
cmd=${PMC_CMD:-"python bench.py --steps 3 --warmup 1 --no-cpu --no-prof --no-extras --full-record /tmp/bench_full_scratch.json"}
export PMC_CMD_TEXT="$cmd"
export DSG_GIT_HEAD=$(cat tools/_head.txt 2>/dev/null || echo unknown)
mkdir -p gpurun_out
timeout 900 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmcm_$tag$PMC_SUFFIX -o p \
  --output-format csv -- $cmd > /dev/null 2> gpurun_out/${tag}_pmc_mfma$PMC_SUFFIX.err
python - "$tag" <<'PY'
import csv, glob, json, os, re, sys, collections
tag = sys.argv[1]
sfx = os.environ.get("PMC_SUFFIX", "")
f = glob.glob(f"/tmp/pmcm_{tag}{sfx}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    if "dsg::" not in r["Kernel_Name"]:
        continue
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        cnt[k] += 1
res = {}
for k, v in acc.items():
    n = cnt[k]
    g = v.get("GRBM_GUI_ACTIVE", 0.0) / n / 8.0  # per-XCD cycles
    m = v.get("SQ_INSTS_MFMA", 0.0) / n
    b = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
    res[k] = {"launches": n, "SQ_INSTS_MFMA": m, "SQ_VALU_MFMA_BUSY_CYCLES": b, "SQ_BUSY_CYCLES": v.get("SQ_BUSY_CYCLES", 0.0) / n,
              "cycles (GRBM_GUI_ACTIVE / 8 XCDs)": g, "mfma_util_issued": (m * 32.0 / (g * 1024.0)) if g else None,
              "mfma_util_counter": (b / (g * 1024.0)) if g else None}
json.dump({"git_head": os.environ.get("DSG_GIT_HEAD", "unknown"), "source": "rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over: " + os.environ.get("PMC_CMD_TEXT", ""),
           "kernels": res}, open(f"gpurun_out/{tag}_pmc_mfma{sfx}.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["cycles (GRBM_GUI_ACTIVE / 8 XCDs)"] * kv[1]["launches"])[:8]:
    print(f'{k[:58]:58s} n={v["launches"]:4d} insts {v["SQ_INSTS_MFMA"]:12.0f} busy {v["SQ_VALU_MFMA_BUSY_CYCLES"]:12.0f} cycles {v["cycles (GRBM_GUI_ACTIVE / 8 XCDs)"]:10.0f} '
          f'util issued {v["mfma_util_issued"] or 0:.3f} counter {v["mfma_util_counter"] or 0:.3f}')
PY
