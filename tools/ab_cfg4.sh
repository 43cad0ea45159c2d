#!/bin/bash
# Same-box sweep of kernel-selection knobs (DSG_TUNING sets) over the configs[3] leg (512x512x4, 6-level attention network, batch 8):
# fp32-equivalent and bf16, interleaved REPS times.  One line per set and repetition.
#   AB_SETS="19=1 20=0 3=4 17=0 27=4" REPS=2 bash tools/ab_cfg4.sh     (the first set should be a no-op baseline)
cd "$GRAFT_REPO_ROOT" || exit 1
export DSG_TESTING=1
num() { grep -o "[0-9.]* ms/step" | head -1; }
for r in $(seq 1 ${REPS:-2}); do
  for v in ${AB_SETS}; do
    a=$(DSG_TUNING=$v python tools/fwd_bench.py cfg4 8 20 fp32 2>/dev/null | tail -1 | num)
    b=$(DSG_TUNING=$v python tools/fwd_bench.py cfg4 8 20 bf16 2>/dev/null | tail -1 | num)
    echo "rep $r tuning $v | cfg4 fp32-eq $a | bf16 $b"
  done
done
