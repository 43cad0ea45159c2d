mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ups_dgrad.py tests/test_gpu_pack_batch.py tests/test_gpu_mixed_train.py -q -x -p no:cacheprovider 2>&1 | tail -15
{
for i in 1 2 3; do
  echo "window4 (default)"; timeout 300 python tools/train_bench.py 128 4 bf16 2>&1 | grep "ms/step"
  echo "full-res 3x3 + 2x2 sums (DSG_UPS_DGRAD_FULLRES=1)"; DSG_UPS_DGRAD_FULLRES=1 timeout 300 python tools/train_bench.py 128 4 bf16 2>&1 | grep "ms/step"
done
for i in 1 2 3; do
  echo "window4 (default)"; timeout 300 python tools/train_bench.py 64 4 fp32 2>&1 | grep "ms/step"
  echo "full-res 3x3 + 2x2 sums (DSG_UPS_DGRAD_FULLRES=1)"; DSG_UPS_DGRAD_FULLRES=1 timeout 300 python tools/train_bench.py 64 4 fp32 2>&1 | grep "ms/step"
done
} > gpurun_out/ups_dgrad_ab.txt 2>&1
cat gpurun_out/ups_dgrad_ab.txt
