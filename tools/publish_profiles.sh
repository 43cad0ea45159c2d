#!/bin/bash
# Copy what tools/collect_r02.sh <tag> left in gpurun_out/ into profiles/ (the tracked, judged copies).
# Usage (in the build container, after the gpurun call returned): bash tools/publish_profiles.sh <tag>
cd "$(dirname "$0")/.."
tag=${1:-r02}
for f in profiles/${tag}_*; do
  b=$(basename "$f")
  [ "$b" = "${tag}_other_runs.txt" ] && continue
  if [ -f "gpurun_out/$b" ]; then cp "gpurun_out/$b" "profiles/$b"; else echo "missing gpurun_out/$b"; fi
done
cat gpurun_out/${tag}_train_fp32_stdout.txt gpurun_out/${tag}_train_bf16_stdout.txt gpurun_out/${tag}_train_b64.txt \
    gpurun_out/${tag}_configs.txt | grep -v amdgpu.ids > profiles/${tag}_other_runs.txt
