#!/bin/bash
# A/B of one tuning key on the headline leg and the bf16 forward, inside one gpurun call:  bash tools/ab.sh <key> <v0> <v1> ...
key=$1; shift
for v in "$@"; do
  DSG_TUNING=$key=$v python bench.py --steps 20 --warmup 10 --no-cpu --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('key $key=$v  fp32-eq', round(r['value'],1), 'img-steps/s', round(r['ms_per_step'],3), 'ms')"
  DSG_TUNING=$key=$v python tools/fwd_bench.py cfg5 64 20 bf16 2>&1 | tail -1
done
