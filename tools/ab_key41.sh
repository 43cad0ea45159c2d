#!/bin/bash
# Same-box A/B: the 16-bit data-gradient convs with the GNB epilogue on 64-cout workgroups, two per CU (dsg_set_tuning key 41 = 1, default)
# against the 128-cout workgroups the plain conv takes (key 41 = 0), bf16 B=128, interleaved.
mkdir -p gpurun_out
{
for i in 1 2 3; do
  for k in 1 0; do
    echo "key 41 = $k"; DSG_TUNING="41=$k" timeout 300 python tools/train_bench.py 128 4 bf16 2>&1 | grep "ms/step"
  done
done
} > gpurun_out/key41_ab.txt 2>&1
cat gpurun_out/key41_ab.txt
