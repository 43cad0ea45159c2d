#!/bin/bash
# A/B of the pre-staged operand images (tuning key 26; threshold key 27) and the early-barrier K loop (key 28) inside one gpurun call
cd "$GRAFT_REPO_ROOT" || exit 1
for v in ${AB_SETS:-"26=0" "26=1" "26=1,27=8" "26=0,28=1" "26=1,28=1"}; do
  DSG_TUNING=$v python bench.py --steps 20 --warmup 10 --no-cpu --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tuning $v  fp32-eq', round(r['value'],1), 'img-steps/s', round(r['ms_per_step'],3), 'ms')"
done
DSG_TUNING=26=0 python bench.py --no-cpu --no-extras --steps 20 --prof-dump gpurun_out/pre0_conv_launches.csv > /dev/null 2>&1
DSG_TUNING=26=1 python bench.py --no-cpu --no-extras --steps 20 --prof-dump gpurun_out/pre1_conv_launches.csv > /dev/null 2>&1
DSG_TUNING=26=0,28=1 python bench.py --no-cpu --no-extras --steps 20 --prof-dump gpurun_out/eb1_conv_launches.csv > /dev/null 2>&1
