#!/bin/bash
# Round 6, late: the fp32 tape changed after tools/collect_r06.sh ran (q/k/v projection and conv_in / conv_out weight gradients on
# the split kernels, attention backward on the matrix cores, residual sums inside the GroupNorm backward): the bench line and the
# fp32 training leg's records again, on one box, plus the same-box A/B of the round-6 routes against round 5's.
#   bash tools/gpu.sh --timeout 1800 -- 'bash tools/collect_r06_late.sh r06'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1
mkdir -p gpurun_out
timeout 1200 python bench.py --full-record gpurun_out/${tag}_bench.json > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
bash tools/prof_cmd.sh ${tag}_train_fp32 python tools/train_bench.py 64 3 fp32 > /dev/null
bash tools/prof_cmd.sh ${tag}_train_bf16 python tools/train_bench.py 128 3 bf16 > /dev/null
PMC_CMD="python tools/train_bench.py 64 1 fp32" bash tools/pmc_bench.sh ${tag} _train_fp32 > gpurun_out/${tag}_pmc_traffic_train_fp32.txt
{ echo "fp32, batch 64 (tools/train_bench.py 64 4 fp32), interleaved on one box";
  for r in 1 2 3; do echo "round-6 fp32 tape"; python tools/train_bench.py 64 4 fp32 2>&1 | tail -1;
    echo "round-5 routes (DSG_F32_TAPE_R5=1 DSG_TUNING=37=0,38=0)"; DSG_F32_TAPE_R5=1 DSG_TUNING="37=0,38=0" python tools/train_bench.py 64 4 fp32 2>&1 | tail -1; done;
  echo "bf16, batch 128 (tools/train_bench.py 128 4 bf16)";
  for r in 1 2 3; do echo "round-6 routes"; python tools/train_bench.py 128 4 bf16 2>&1 | tail -1;
    echo "round-5 routes (DSG_TUNING=37=0 DSG_W16_SAMPLER=0)"; DSG_TUNING="37=0" DSG_W16_SAMPLER=0 python tools/train_bench.py 128 4 bf16 2>&1 | tail -1; done; } > gpurun_out/${tag}_routes_ab.txt 2>&1
ls -la gpurun_out | grep ${tag}_ | head -40
