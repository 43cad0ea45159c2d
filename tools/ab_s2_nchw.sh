#!/bin/bash
# Same-box A/B: the fp32 tape's down-sampler convs on the space-to-depth kernel (dsg_set_tuning key 40 = 1, default) against the exact
# f32 MFMA kernel (key 40 = 0), fp32 B=64, interleaved.
mkdir -p gpurun_out
{
for i in 1 2 3; do
  echo "space-to-depth kernel (default)"; timeout 300 python tools/train_bench.py 64 4 fp32 2>&1 | grep "ms/step"
  echo "exact f32 kernel (DSG_TUNING=40=0)"; DSG_TUNING="40=0" timeout 300 python tools/train_bench.py 64 4 fp32 2>&1 | grep "ms/step"
done
} > gpurun_out/s2_nchw_ab.txt 2>&1
cat gpurun_out/s2_nchw_ab.txt
