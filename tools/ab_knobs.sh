#!/bin/bash
# Same-box sweep of kernel-selection knobs (DSG_TUNING sets) over the sampling legs: headline (configs[1], B=16), bf16 forward
# (configs[4] network, B=64), batch 1 and batch 5 of the reference's sampling calls.  One line per set.
#   AB_SETS="17=0 20=0 18=0" bash tools/ab_knobs.sh      (the first set should be a no-op, e.g. 19=1, as the baseline)
cd "$GRAFT_REPO_ROOT" || exit 1
export DSG_TESTING=1
num() { grep -o "[0-9.]* ms/step" | head -1; }
for v in ${AB_SETS}; do
  h=$(DSG_TUNING=$v python bench.py --steps 20 --warmup 5 --no-cpu --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['ms_per_step'],3))")
  b=$(DSG_TUNING=$v python tools/fwd_bench.py cfg5 64 15 bf16 2>/dev/null | tail -1 | num)
  s1=$(DSG_TUNING=$v python tools/fwd_bench.py default3 1 100 fp32 2>/dev/null | tail -1 | num)
  s5=$(DSG_TUNING=$v python tools/fwd_bench.py default3 5 60 fp32 2>/dev/null | tail -1 | num)
  echo "tuning $v | headline $h ms | bf16 fwd $b | B=1 $s1 | B=5 $s5"
done
