#!/bin/bash
# SQ counters of the 16-bit weight-gradient kernels over tools/wgrad16_one.py (one layer shape, wide and 64x64 workgroups).
# Usage (on the GPU box): bash tools/pmc_wgrad16.sh <tag> <cin> <cout> <h> <w> [batch]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp DSG_TESTING=1
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_${tag}_$i -o p --output-format csv -- python tools/wgrad16_one.py "$@" > /dev/null 2>$out/err_$i.txt
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py "$f" conv_wgrad16 | tail -3 > $out/set_$i.txt
done
cat $out/set_*.txt
