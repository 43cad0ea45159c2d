#!/bin/bash
# Copy what tools/collect_r06.sh <tag> left in gpurun_out/ into profiles/ (the tracked, judged copies).
# Usage (in the build container, after the gpurun call returned): bash tools/publish_profiles.sh <tag>
cd "$(dirname "$0")/.."
tag=${1:-r06}
for f in bench.json bench_line.json kernel_stats.csv conv_launches.csv conv_launches_unfused.csv ab.txt pmc_traffic.json pmc_mfma.json \
         bf16_fwd_kernel_stats.csv bf16_fwd_conv_launches.csv pmc_traffic_bf16.json pmc_mfma_bf16.json \
         train_fp32_kernel_stats.csv train_bf16_kernel_stats.csv pmc_traffic_train_fp32.json pmc_traffic_train_bf16.json \
         small_batch_b1_kernel_stats.csv small_batch_b5_kernel_stats.csv winograd_ablation.txt \
         cfg4_kernel_stats.csv cfg4_conv_launches.csv pmc_traffic_cfg4.json pmc_mfma_cfg4.json \
         gnb_ab.txt train_bf16_gnb_off_kernel_stats.csv cpu_threads_sweep.txt; do
  if [ -f "gpurun_out/${tag}_$f" ]; then cp "gpurun_out/${tag}_$f" "profiles/${tag}_$f"; else echo "missing gpurun_out/${tag}_$f"; fi
done
cat gpurun_out/${tag}_cfg4_stdout.txt gpurun_out/${tag}_train_fp32_stdout.txt gpurun_out/${tag}_train_bf16_stdout.txt gpurun_out/${tag}_other_runs.txt \
    gpurun_out/${tag}_small_batch_b1_stdout.txt gpurun_out/${tag}_small_batch_b5_stdout.txt 2>/dev/null | grep -v amdgpu.ids > profiles/${tag}_other_runs.txt
