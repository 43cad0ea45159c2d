"""Per-layer minimum over the settings of tools/geometry_sweep.sh: records are aligned by their position in the cumulative
algorithmic-FLOP stream of a step (a fused shortcut record = its two unfused records), times of the 2nd..last step averaged."""
import csv, glob, sys, collections
tag = sys.argv[1]
runs = {}
for f in sorted(glob.glob(f"gpurun_out/{tag}_s*.csv")):
    rows = [(int(r["kernel_class"]), float(r["alg_flops"]), float(r["ms"])) for r in csv.DictReader(open(f))]
    idx = [i for i, r in enumerate(rows) if r[0] == 11]          # conv_in starts a step
    steps = [rows[a:b] for a, b in zip(idx[1:-1], idx[2:])]      # skip the first
    n = len(steps[0])
    steps = [s for s in steps if len(s) == n]
    avg = [(steps[0][i][0], steps[0][i][1], sum(s[i][2] for s in steps) / len(steps)) for i in range(n)]
    runs[f.split(tag + "_")[1][:-4]] = avg
base = runs[sorted(runs)[0]]
tot = sum(r[1] for r in base)
# segment boundaries = cumulative flops of the base run; a run's time in a segment = sum of its records whose midpoint falls inside
def seg_times(run, bounds):
    out = [0.0] * (len(bounds) - 1)
    c = 0.0
    for _, fl, ms in run:
        mid = c + fl / 2
        c += fl
        for k in range(len(bounds) - 1):
            if bounds[k] <= mid < bounds[k + 1]:
                out[k] += ms
                break
    return out
# coarse segments: merge base records so that every run's boundaries align (a fused record spans two unfused ones)
cums = {}
for name, run in runs.items():
    c, s = 0.0, set()
    for _, fl, _ in run:
        c += fl
        s.add(round(c / 1e6))
    cums[name] = s
common = sorted(set.intersection(*cums.values()))
bounds = [0.0] + [c * 1e6 + 1 for c in common]
table = {name: seg_times(run, bounds) for name, run in runs.items()}
names = sorted(runs)
print("segments", len(bounds) - 1)
print("setting totals (ms, conv records only):", {n: round(sum(table[n]), 3) for n in names})
best = [min(table[n][k] for n in names) for k in range(len(bounds) - 1)]
d0 = names[0]
print(f"default {sum(table[d0]):.3f} ms  per-layer best {sum(best):.3f} ms  ({100 * (1 - sum(best) / sum(table[d0])):.1f} % less)")
wins = collections.Counter()
for k in range(len(bounds) - 1):
    w = min(names, key=lambda n: table[n][k])
    if table[w][k] < 0.97 * table[d0][k]:
        wins[w] += 1
        print(f"  seg {k:2d} flops {(bounds[k+1]-bounds[k])/1e9:7.2f} G  default {table[d0][k]*1e3:7.1f} us  best {table[w][k]*1e3:7.1f} us  {w}")
print(wins)
