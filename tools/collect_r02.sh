#!/bin/bash
# Everything the round-2 numbers in DESIGN.md / profiles/ come from, in one call on the GPU box:
#   bash tools/collect_r02.sh <tag>      -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1
mkdir -p gpurun_out "$(dirname gpurun_out/${tag}_x)" "$(dirname /tmp/prof_${tag}_x)"
# 1. the bench line as the driver runs it (extras and CPU baseline on)
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
# 2. rocprofv3 kernel-trace summary of the headline leg; per-launch HIP-event records
bash tools/prof_bench.sh ${tag} --steps 20 > /dev/null
python bench.py --no-cpu --no-extras --steps 20 --prof-dump gpurun_out/${tag}_conv_launches.csv > /dev/null 2>&1
# 3. PMC passes of the headline leg (HBM traffic; matrix-pipe utilisation)
bash tools/pmc_bench.sh ${tag} > gpurun_out/${tag}_pmc_traffic.txt
bash tools/pmc_mfma.sh ${tag} > gpurun_out/${tag}_pmc_mfma.txt
# 4. the mixed-precision forward (configs[4] network, bf16, batch 64): kernel stats, HBM traffic, matrix-pipe utilisation
bash tools/prof_cmd.sh ${tag}_bf16_fwd python tools/fwd_bench.py cfg5 64 20 bf16 > /dev/null
PMC_CMD="python tools/fwd_bench.py cfg5 64 3 bf16" bash tools/pmc_bench.sh ${tag} _bf16 > gpurun_out/${tag}_pmc_traffic_bf16.txt
PMC_CMD="python tools/fwd_bench.py cfg5 64 3 bf16" bash tools/pmc_mfma.sh ${tag} _bf16 > gpurun_out/${tag}_pmc_mfma_bf16.txt
PROF_DUMP=gpurun_out/${tag}_bf16_fwd_conv_launches.csv python tools/fwd_bench.py cfg5 64 4 bf16 > /dev/null 2>&1
# 4b. where a workgroup's time goes (instrumented builds of the conv kernels, tools/build_variant.sh timing*) and the
#     matrix-pipe issue model (tools/probes/probe_mfma.hip)
if [ -f drivescenegen_amd/lib/libdsg_timing.so ]; then
  BLOCKED=1 DSG_LIB_PATH=drivescenegen_amd/lib/libdsg_timing.so python tools/h2_timing.py > gpurun_out/${tag}_h2_timing_fp32.txt 2>&1
  DSG_TUNING=20=0 BLOCKED=1 DSG_LIB_PATH=drivescenegen_amd/lib/libdsg_timing.so python tools/h2_timing.py > gpurun_out/${tag}_h2_timing_fp32_one_wg_per_cu.txt 2>&1
fi
if [ -f drivescenegen_amd/lib/libdsg_timing16.so ]; then
  BLOCKED=1 DTYPE=bf16 BSCALE=4 DSG_LIB_PATH=drivescenegen_amd/lib/libdsg_timing16.so python tools/h2_timing.py > gpurun_out/${tag}_h2_timing_bf16.txt 2>&1
fi
[ -x tools/_build/probe_mfma ] && tools/_build/probe_mfma > gpurun_out/${tag}_probe_mfma.txt 2>&1
# 4c. clocks and power while the headline leg runs (the matrix-core loop is power-limited: see DESIGN.md)
python bench.py --no-cpu --no-extras --steps 2500 > /dev/null 2>&1 &
bpid=$!
for t in 1 2 3 4 5 6; do sleep 8; echo "== t=$((8 * t)) s"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|mclk|Power|fclk" ; done > gpurun_out/${tag}_clocks_power.txt 2>&1
wait $bpid
# 5. training steps
bash tools/prof_cmd.sh ${tag}_train_fp32 python tools/train_bench.py 16 3 fp32 > /dev/null
bash tools/prof_cmd.sh ${tag}_train_bf16 python tools/train_bench.py 32 3 bf16 > /dev/null
python tools/train_bench.py 64 3 fp32 > gpurun_out/${tag}_train_b64.txt 2>&1
python tools/train_bench.py 64 3 bf16 >> gpurun_out/${tag}_train_b64.txt 2>&1
python tools/train_bench.py 128 3 bf16 >> gpurun_out/${tag}_train_b64.txt 2>&1
python tools/train_bench.py 32 3 fp16 >> gpurun_out/${tag}_train_b64.txt 2>&1
# 6. the other configs at full size
python tools/fwd_bench.py cfg4 8 20 fp32 > gpurun_out/${tag}_configs.txt 2>&1
python tools/fwd_bench.py cfg4 8 20 bf16 >> gpurun_out/${tag}_configs.txt 2>&1
python tools/fwd_bench.py default3 1 50 fp32 >> gpurun_out/${tag}_configs.txt 2>&1
python tools/fwd_bench.py default3 5 50 fp32 >> gpurun_out/${tag}_configs.txt 2>&1
python tools/fwd_bench.py cfg5 128 10 bf16 >> gpurun_out/${tag}_configs.txt 2>&1
ls -la gpurun_out | grep ${tag} | head -40
