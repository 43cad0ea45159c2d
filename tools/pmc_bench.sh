#!/bin/bash
# HBM traffic of the bench's kernels from the memory-side L2 counters, one timeout 900 rocprofv3 --pmc pass per counter
# (kernel-trace only).  Writes gpurun_out/<tag>_pmc_traffic.json: per kernel, launches and mean bytes per launch,
# FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950 -- checked here
# on a 1x1 conv with a known read size (805.3 MB read: FETCH_SIZE reported 394,948 KiB = 0.502 of it; WRITE_SIZE
# 268,288 KiB for a 268.4 MB output = 1.02).
# Usage (on the GPU box): bash tools/pmc_bench.sh <tag> [suffix]      (PMC_CMD="python tools/fwd_bench.py ..." profiles
# another command; suffix names the output gpurun_out/<tag>_pmc_traffic<suffix>.json, e.g. _bf16)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1
export PMC_SUFFIX=${2:-}
cmd=${PMC_CMD:-"python bench.py --steps 3 --warmup 1 --no-cpu --no-prof --no-extras --full-record /tmp/bench_full_scratch.json"}
export PMC_CMD_TEXT="$cmd"
export DSG_GIT_HEAD=$(cat tools/_head.txt 2>/dev/null || echo unknown)
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmcb_${tag}${PMC_SUFFIX}_$c -o p --output-format csv -- \
    $cmd > /dev/null 2> gpurun_out/${tag}_pmc_$c$PMC_SUFFIX.err
done
python - "$tag" <<'PY'
import csv, glob, json, os, re, sys, collections
tag = sys.argv[1]
sfx = os.environ.get("PMC_SUFFIX", "")
out = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": collections.Counter()})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmcb_{tag}{sfx}_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if "dsg::" not in r["Kernel_Name"] or r["Counter_Name"] != c:
            continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        out[k][c] += float(r["Counter_Value"])
        out[k]["n"][c] += 1
res = {}
for k, v in out.items():
    n = max(v["n"]["FETCH_SIZE"], 1)
    res[k] = {"launches": n,
              "fetch_bytes_per_launch": 2.0 * 1024.0 * v["FETCH_SIZE"] / n,   # KiB, x2 (gfx950 correction)
              "write_bytes_per_launch": 1024.0 * v["WRITE_SIZE"] / max(v["n"]["WRITE_SIZE"], 1)}
    res[k]["hbm_bytes_per_launch"] = res[k]["fetch_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
json.dump({"git_head": os.environ.get("DSG_GIT_HEAD", "unknown"),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over: " + os.environ.get("PMC_CMD_TEXT", ""),
           "correction": "FETCH_SIZE x2 (gfx950), KiB units; WRITE_SIZE as reported", "kernels": res},
          open(f"gpurun_out/{tag}_pmc_traffic{sfx}.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(f'{k[:60]:60s} n={v["launches"]:5d} fetch {v["fetch_bytes_per_launch"]/1e6:9.1f} MB write {v["write_bytes_per_launch"]/1e6:9.1f} MB')
PY
