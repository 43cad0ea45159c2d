#!/bin/bash
# A/B of alternative library builds (DSG_LIB_PATH) inside one gpurun call: headline leg, bf16 forward, both training steps
#   LIBS="libdsg.so libdsg_noslp.so" REPS=2 bash tools/ab_lib.sh
cd "$GRAFT_REPO_ROOT" || exit 1
for r in $(seq 1 ${REPS:-2}); do
for l in ${LIBS}; do
  export DSG_LIB_PATH=$GRAFT_REPO_ROOT/drivescenegen_amd/lib/$l
  python bench.py --steps 30 --warmup 10 --no-cpu --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l  fp32-eq', round(r['value'],1), 'img-steps/s', round(r['ms_per_step'],3), 'ms')"
  echo "$l $(python tools/fwd_bench.py cfg5 64 20 bf16 2>&1 | tail -1)"
  if [ -z "$NO_TRAIN" ]; then
  echo "$l $(python tools/train_bench.py 16 3 fp32 2>&1 | tail -1)"
  echo "$l $(python tools/train_bench.py 32 3 bf16 2>&1 | tail -1)"
  fi
done
done
