#!/bin/bash
# A/B of DSG_TUNING sets on the headline leg inside one gpurun call, REPS passes over the sets (interleaved: boxes drift)
#   AB_SETS="26=0 26=1 26=1,27=8" REPS=2 bash tools/ab_sets.sh
cd "$GRAFT_REPO_ROOT" || exit 1
for r in $(seq 1 ${REPS:-2}); do
for v in ${AB_SETS}; do
  DSG_TUNING=$v python bench.py --steps ${AB_STEPS:-30} --warmup 10 --no-cpu --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tuning $v  fp32-eq', round(r['value'],1), 'img-steps/s', round(r['ms_per_step'],3), 'ms')"
done
done
