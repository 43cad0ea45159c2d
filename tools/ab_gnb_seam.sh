#!/bin/bash
# Same-box A/B: GroupNorm-backward epilogue also for concat convs whose x tensors meet inside a channel tile (default) against
# key 37 = 3 (round 6 first rule: such calls run the statistics pass), bf16 B=128 and fp32 B=64, interleaved.
mkdir -p gpurun_out
{
for i in 1 2 3; do
  echo "x tensors meeting inside a 128-cout tile keep the epilogue (default)"; timeout 300 python tools/train_bench.py 128 4 bf16 2>&1 | grep "ms/step"
  echo "such calls lose it to the statistics pass (DSG_TUNING=37=3)"; DSG_TUNING="37=3" timeout 300 python tools/train_bench.py 128 4 bf16 2>&1 | grep "ms/step"
done
for i in 1 2; do
  echo "fp32: default"; timeout 300 python tools/train_bench.py 64 4 fp32 2>&1 | grep "ms/step"
  echo "fp32: DSG_TUNING=37=3"; DSG_TUNING="37=3" timeout 300 python tools/train_bench.py 64 4 fp32 2>&1 | grep "ms/step"
done
} > gpurun_out/gnb_seam_ab.txt 2>&1
cat gpurun_out/gnb_seam_ab.txt
