#!/bin/bash
# timeout 900 rocprofv3 --kernel-trace --stats over the bench.py command line; writes a compact per-kernel summary
# (CSV: kernel, calls, total_ms, avg_us, pct) to gpurun_out/<tag>_kernel_stats.csv.
# Usage (on the GPU box): bash tools/prof_bench.sh <tag> [bench.py flags...]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p --output-format csv -- python bench.py --no-cpu --no-extras --full-record /tmp/bench_full_scratch.json "$@" \
  > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_rocprof.err
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" gpurun_out/${tag}_kernel_stats.csv <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
out = csv.writer(open(sys.argv[2], "w"))
out.writerow(["kernel", "calls", "total_ms", "avg_us", "pct"])
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"])[-90:]
    out.writerow([name, r["Calls"], f'{float(r["TotalDurationNs"]) / 1e6:.3f}', f'{float(r["AverageNs"]) / 1e3:.2f}',
                  r["Percentage"]])
PY
head -25 gpurun_out/${tag}_kernel_stats.csv
