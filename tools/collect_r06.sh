#!/bin/bash
# Everything the round-6 numbers in DESIGN.md / profiles/ come from, in one call on the GPU box (start it through tools/gpu.sh so
# that the counter files carry the git head):
#   bash tools/collect_r06.sh <tag>      -> gpurun_out/<tag>_*  (tools/publish_profiles.sh <tag> copies what is judged into profiles/)
# Training legs run at the BASELINE configurations' own batch sizes (configs[2]: 64, configs[4]: 128); configs[3] (512^2, batch 8)
# gets its own kernel-trace, per-launch and HBM-traffic records.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1
mkdir -p gpurun_out
# (every command runs under `timeout`: a GPU fault under rocprofv3 once left the profiler hanging for 43 minutes of box time)
# 1. the bench line as the driver runs it (extras and CPU baseline on)
# (stdout = the compact line the driver parses; the full record -- every class row of every leg -- goes to its own file)
timeout 1200 python bench.py --full-record gpurun_out/${tag}_bench.json > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
# 2. headline: rocprofv3 kernel-trace summary, per-launch HIP-event records, PMC passes (HBM traffic; matrix-pipe utilisation)
bash tools/prof_bench.sh ${tag} --steps 20 > /dev/null
python bench.py --no-cpu --no-extras --steps 20 --prof-dump gpurun_out/${tag}_conv_launches.csv --full-record /tmp/bench_full_scratch.json > /dev/null 2>&1
bash tools/pmc_bench.sh ${tag} > gpurun_out/${tag}_pmc_traffic.txt
bash tools/pmc_mfma.sh ${tag} > gpurun_out/${tag}_pmc_mfma.txt
# 3. configs[3]: 512x512x4, the 6-level attention network, batch 8, fp32-equivalent (and bf16 for reference)
bash tools/prof_cmd.sh ${tag}_cfg4 python tools/fwd_bench.py cfg4 8 20 fp32 > /dev/null
PROF_DUMP=gpurun_out/${tag}_cfg4_conv_launches.csv python tools/fwd_bench.py cfg4 8 4 fp32 > /dev/null 2>&1
PMC_CMD="python tools/fwd_bench.py cfg4 8 3 fp32" bash tools/pmc_bench.sh ${tag} _cfg4 > gpurun_out/${tag}_pmc_traffic_cfg4.txt
PMC_CMD="python tools/fwd_bench.py cfg4 8 3 fp32" bash tools/pmc_mfma.sh ${tag} _cfg4 > gpurun_out/${tag}_pmc_mfma_cfg4.txt
# 4. the mixed-precision forward (configs[4] network, bf16, batch 64)
bash tools/prof_cmd.sh ${tag}_bf16_fwd python tools/fwd_bench.py cfg5 64 20 bf16 > /dev/null
PMC_CMD="python tools/fwd_bench.py cfg5 64 3 bf16" bash tools/pmc_bench.sh ${tag} _bf16 > gpurun_out/${tag}_pmc_traffic_bf16.txt
PMC_CMD="python tools/fwd_bench.py cfg5 64 3 bf16" bash tools/pmc_mfma.sh ${tag} _bf16 > gpurun_out/${tag}_pmc_mfma_bf16.txt
PROF_DUMP=gpurun_out/${tag}_bf16_fwd_conv_launches.csv python tools/fwd_bench.py cfg5 64 4 bf16 > /dev/null 2>&1
# 5. training steps at the configs' own batch: kernel stats, HBM traffic of the kernels the training records' roofline objects name
bash tools/prof_cmd.sh ${tag}_train_fp32 python tools/train_bench.py 64 3 fp32 > /dev/null
bash tools/prof_cmd.sh ${tag}_train_bf16 python tools/train_bench.py 128 3 bf16 > /dev/null
PMC_CMD="python tools/train_bench.py 64 1 fp32" bash tools/pmc_bench.sh ${tag} _train_fp32 > gpurun_out/${tag}_pmc_traffic_train_fp32.txt
PMC_CMD="python tools/train_bench.py 128 1 bf16" bash tools/pmc_bench.sh ${tag} _train_bf16 > gpurun_out/${tag}_pmc_traffic_train_bf16.txt
# (GroupNorm-backward statistics from the data-gradient convs' epilogues, tuning key 37: same-box A/B, interleaved)
{ for k in 1 0 1 0; do echo "key 37 = $k"; DSG_TUNING="37=$k" python tools/train_bench.py 128 4 bf16 | tail -1; done
  for k in 1 0 1 0; do echo "key 37 = $k"; DSG_TUNING="37=$k" python tools/train_bench.py 64 4 fp32 | tail -1; done; } > gpurun_out/${tag}_gnb_ab.txt 2>&1
DSG_TUNING="37=0" bash tools/prof_cmd.sh ${tag}_train_bf16_gnb_off python tools/train_bench.py 128 3 bf16 > /dev/null
python tools/cpu_threads_sweep.py 8 16 32 64 128 > gpurun_out/${tag}_cpu_threads_sweep.txt 2>&1
{ python tools/train_bench.py 16 3 fp32; DSG_TUNING=31=0 python tools/train_bench.py 64 3 fp32; python tools/train_bench.py 64 3 fp32; python tools/train_bench.py 32 3 bf16;
  python tools/train_bench.py 128 3 bf16; python tools/train_bench.py 14 3 fp16; python tools/ref_point_probe.py 10; } > gpurun_out/${tag}_other_runs.txt 2>&1
python tools/whole_call_probe.py 750 >> gpurun_out/${tag}_other_runs.txt 2>&1
# 6. the other sampling legs
{ python tools/fwd_bench.py cfg4 8 20 fp32; python tools/fwd_bench.py cfg4 8 20 bf16; python tools/fwd_bench.py default3 1 50 fp32; python tools/fwd_bench.py default3 5 50 fp32;
  python tools/fwd_bench.py cfg5 128 10 bf16; } >> gpurun_out/${tag}_other_runs.txt 2>&1
bash tools/prof_cmd.sh ${tag}_small_batch_b1 python tools/fwd_bench.py default3 1 100 fp32 > /dev/null
bash tools/prof_cmd.sh ${tag}_small_batch_b5 python tools/fwd_bench.py default3 5 100 fp32 > /dev/null
ls -la gpurun_out | grep ${tag} | head -80
