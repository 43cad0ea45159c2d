#!/bin/bash
# Round 6, final: the training tapes changed after tools/collect_r06_late.sh ran (up-sampler convs: data gradient as one 4x4 stride-2
# window, weight gradient folded onto x's own map; GroupNorm-backward epilogue for the 64 + 64 concat convs and on 64-cout workgroups;
# the fp32 tape's down-samplers on the space-to-depth kernel): the bench line, the
# training legs' kernel / HBM-traffic records and the same-box A/B of the round's training routes against round 5's again, on one
# box, then the whole GPU suite alone and next to a mixed load.
#   bash tools/gpu.sh --timeout 3000 -- 'bash tools/collect_r06_final.sh r06'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1
mkdir -p gpurun_out
timeout 1200 python bench.py --full-record gpurun_out/${tag}_bench.json > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
bash tools/prof_cmd.sh ${tag}_train_fp32 python tools/train_bench.py 64 3 fp32 > /dev/null
bash tools/prof_cmd.sh ${tag}_train_bf16 python tools/train_bench.py 128 3 bf16 > /dev/null
PMC_CMD="python tools/train_bench.py 64 1 fp32" bash tools/pmc_bench.sh ${tag} _train_fp32 > gpurun_out/${tag}_pmc_traffic_train_fp32.txt
PMC_CMD="python tools/train_bench.py 128 1 bf16" bash tools/pmc_bench.sh ${tag} _train_bf16 > gpurun_out/${tag}_pmc_traffic_train_bf16.txt
PMC_CMD="python tools/train_bench.py 64 1 fp32" bash tools/pmc_mfma.sh ${tag} _train_fp32 > gpurun_out/${tag}_pmc_mfma_train_fp32.txt
PMC_CMD="python tools/train_bench.py 128 1 bf16" bash tools/pmc_mfma.sh ${tag} _train_bf16 > gpurun_out/${tag}_pmc_mfma_train_bf16.txt
R5F="DSG_F32_TAPE_R5=1 DSG_UPS_DGRAD_FULLRES=1 DSG_TUNING=37=0,38=0,40=0"
R5B="DSG_UPS_DGRAD_FULLRES=1 DSG_W16_SAMPLER=0 DSG_TUNING=37=0,39=0"
{ echo "fp32, batch 64 (tools/train_bench.py 64 4 fp32), interleaved on one box";
  for r in 1 2 3; do echo "round-6 fp32 tape"; timeout 300 python tools/train_bench.py 64 4 fp32 2>&1 | tail -1;
    echo "round-5 routes ($R5F)"; env $R5F timeout 300 python tools/train_bench.py 64 4 fp32 2>&1 | tail -1; done;
  echo "bf16, batch 128 (tools/train_bench.py 128 4 bf16)";
  for r in 1 2 3; do echo "round-6 routes"; timeout 300 python tools/train_bench.py 128 4 bf16 2>&1 | tail -1;
    echo "round-5 routes ($R5B)"; env $R5B timeout 300 python tools/train_bench.py 128 4 bf16 2>&1 | tail -1; done; } > gpurun_out/${tag}_routes_ab.txt 2>&1
(timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4) > gpurun_out/${tag}_suite_alone.txt
LOAD=mix timeout 1300 bash tools/suite_under_load.sh > gpurun_out/${tag}_suite_under_load_mix.txt 2>&1
ls -la gpurun_out | grep ${tag}_ | head -40
tail -2 gpurun_out/${tag}_suite_alone.txt; tail -2 gpurun_out/${tag}_suite_under_load_mix.txt; cat gpurun_out/${tag}_bench_line.json | cut -c1-600
