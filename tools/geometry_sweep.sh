#!/bin/bash
# Per-launch HIP-event records of a forward under several global kernel-selection settings (DSG_TUNING): what a per-layer
# choice of tile geometry could buy at a batch size (the per-launch minimum over the settings = an autotuner's upper bound).
#   bash tools/geometry_sweep.sh <cfg> <batch> <tag>      -> gpurun_out/<tag>_<setting>.csv + one summary line per setting
cd "$GRAFT_REPO_ROOT" || exit 1
cfg=$1; b=$2; tag=$3
mkdir -p gpurun_out
i=0
for s in "" "3=2" "3=4" "17=0" "19=0" "20=0" "16=1" "34=0" "3=2,17=0" "3=2,20=0" "3=4,19=0" "17=0,19=0" "23=0"; do
  name=$(echo "s$i-$s" | tr '=,' '__')
  if [ -z "$s" ]; then env=""; else env="DSG_TUNING=$s"; fi
  line=$(env $env PROF_DUMP=gpurun_out/${tag}_$name.csv python tools/fwd_bench.py $cfg $b 40 fp32 2>/dev/null | tail -1)
  echo "$name: $line"
  i=$((i+1))
done
