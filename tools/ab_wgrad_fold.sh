#!/bin/bash
# Same-box A/B of the folded weight gradient of Upsample2D's conv (dsg_set_tuning key 39) on the bf16 training step, interleaved.
#   bash tools/gpu.sh --timeout 1500 -- 'bash tools/ab_wgrad_fold.sh'
mkdir -p gpurun_out
{
for i in 1 2 3; do
  echo "folded (key 39 = 1, default)"; timeout 300 python tools/train_bench.py 128 4 bf16 2>&1 | grep "ms/step"

  echo "nine taps at full resolution (DSG_TUNING=39=0)"; DSG_TUNING="39=0" timeout 300 python tools/train_bench.py 128 4 bf16 2>&1 | grep "ms/step"
done
} > gpurun_out/wgrad_fold_ab.txt 2>&1
cat gpurun_out/wgrad_fold_ab.txt
