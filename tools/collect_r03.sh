#!/bin/bash
# Everything the round-3 numbers in DESIGN.md / profiles/ come from, in one call on the GPU box:
#   bash tools/collect_r03.sh <tag>      -> gpurun_out/<tag>_*  (tools/publish_profiles.sh copies what is judged into profiles/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1
mkdir -p gpurun_out
# 1. the bench line as the driver runs it (extras and CPU baseline on)
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
# 2. rocprofv3 kernel-trace summary of the headline leg; per-launch HIP-event records; the same with the shortcut fusion off
bash tools/prof_bench.sh ${tag} --steps 20 > /dev/null
python bench.py --no-cpu --no-extras --steps 20 --prof-dump gpurun_out/${tag}_conv_launches.csv > /dev/null 2>&1
DSG_TUNING=23=0 python bench.py --no-cpu --no-extras --steps 20 --prof-dump gpurun_out/${tag}_conv_launches_unfused.csv 2>/dev/null | python -c "
import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shortcut fusion OFF (DSG_TUNING=23=0):', r['value'], 'image-steps/s', r['ms_per_step'], 'ms/step')" > gpurun_out/${tag}_ab.txt
python bench.py --no-cpu --no-extras --steps 20 2>/dev/null | python -c "
import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shortcut fusion ON  (default)        :', r['value'], 'image-steps/s', r['ms_per_step'], 'ms/step')" >> gpurun_out/${tag}_ab.txt
for v in 0 1; do echo "bf16 forward B=64, shortcut fusion $v:" >> gpurun_out/${tag}_ab.txt; DSG_TUNING=23=$v python tools/fwd_bench.py cfg5 64 20 bf16 2>&1 | tail -1 >> gpurun_out/${tag}_ab.txt; done
for v in 0 1; do echo "blocked attention (key 25) $v:" >> gpurun_out/${tag}_ab.txt; DSG_TUNING=25=$v python tools/fwd_bench.py cfg5 64 20 bf16 2>&1 | tail -1 >> gpurun_out/${tag}_ab.txt; done
# 3. PMC passes of the headline leg (HBM traffic; matrix-pipe utilisation)
bash tools/pmc_bench.sh ${tag} > gpurun_out/${tag}_pmc_traffic.txt
bash tools/pmc_mfma.sh ${tag} > gpurun_out/${tag}_pmc_mfma.txt
# 4. the mixed-precision forward (configs[4] network, bf16, batch 64)
bash tools/prof_cmd.sh ${tag}_bf16_fwd python tools/fwd_bench.py cfg5 64 20 bf16 > /dev/null
PMC_CMD="python tools/fwd_bench.py cfg5 64 3 bf16" bash tools/pmc_bench.sh ${tag} _bf16 > gpurun_out/${tag}_pmc_traffic_bf16.txt
PMC_CMD="python tools/fwd_bench.py cfg5 64 3 bf16" bash tools/pmc_mfma.sh ${tag} _bf16 > gpurun_out/${tag}_pmc_mfma_bf16.txt
PROF_DUMP=gpurun_out/${tag}_bf16_fwd_conv_launches.csv python tools/fwd_bench.py cfg5 64 4 bf16 > /dev/null 2>&1
# 5. training steps: kernel stats, HBM traffic of the kernels the training records' roofline objects name
bash tools/prof_cmd.sh ${tag}_train_fp32 python tools/train_bench.py 16 3 fp32 > /dev/null
bash tools/prof_cmd.sh ${tag}_train_bf16 python tools/train_bench.py 32 3 bf16 > /dev/null
PMC_CMD="python tools/train_bench.py 16 1 fp32" bash tools/pmc_bench.sh ${tag} _train_fp32 > gpurun_out/${tag}_pmc_traffic_train_fp32.txt
PMC_CMD="python tools/train_bench.py 32 1 bf16" bash tools/pmc_bench.sh ${tag} _train_bf16 > gpurun_out/${tag}_pmc_traffic_train_bf16.txt
{ python tools/train_bench.py 64 3 fp32; python tools/train_bench.py 64 3 bf16; python tools/train_bench.py 128 3 bf16; python tools/train_bench.py 14 3 fp16; python tools/train_bench.py 32 3 fp16; } > gpurun_out/${tag}_other_runs.txt 2>&1
# 6. the other configs at full size; the reference's sampling batches
{ python tools/fwd_bench.py cfg4 8 20 fp32; python tools/fwd_bench.py cfg4 8 20 bf16; python tools/fwd_bench.py default3 1 50 fp32; python tools/fwd_bench.py default3 5 50 fp32; python tools/fwd_bench.py cfg5 128 10 bf16; } >> gpurun_out/${tag}_other_runs.txt 2>&1
bash tools/prof_cmd.sh ${tag}_small_batch_b1 python tools/fwd_bench.py default3 1 100 fp32 > /dev/null
bash tools/prof_cmd.sh ${tag}_small_batch_b5 python tools/fwd_bench.py default3 5 100 fp32 > /dev/null
ls -la gpurun_out | grep ${tag} | head -60
