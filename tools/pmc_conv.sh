#!/bin/bash
# PMC pass over tools/conv_bench.py for one layer shape (separate passes per counter set; kernel-trace only).
# Usage (on the GPU box): bash tools/pmc_conv.sh <ONLY-substring> <DSG_EXP> <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
only=$1; exp=$2; tag=$3
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  H2=1 ONLY="$only" rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_${tag}_$i -o p --output-format csv -- python tools/conv_bench.py 3 > /dev/null 2>$out/err_$i.txt
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py "$f" conv_h2 | tail -2 > $out/set_$i.txt
done
cat $out/set_*.txt
