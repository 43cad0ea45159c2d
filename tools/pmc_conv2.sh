#!/bin/bash
# L2-side counters for one conv_bench.py shape.  Usage: bash tools/pmc_conv2.sh <ONLY-substring> <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
only=$1; tag=$2
out=gpurun_out/pmc2_$tag; mkdir -p $out
i=0
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  H2=1 ONLY="$only" rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc2_${tag}_$i -o p --output-format csv -- python tools/conv_bench.py 3 > /dev/null 2>$out/err_$i.txt
  f=$(find /tmp/pmc2_${tag}_$i -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py "$f" conv_h2 | tail -2 > $out/set_$i.txt
  python tools/pmc_summary.py "$f" conv_h2 | head -1 >> $out/set_$i.txt
done
cat $out/set_*.txt | cut -c1-200
