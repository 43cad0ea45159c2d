#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
only=$1; tag=$2
H2=1 ONLY="$only" rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH --kernel-trace -d /tmp/pmc3_$tag -o p --output-format csv -- python tools/conv_bench.py 3 > /dev/null 2>/tmp/err_$tag.txt
f=$(find /tmp/pmc3_$tag -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py "$f" conv_h2 | head -1 | cut -c1-220; python tools/pmc_summary.py "$f" conv_h2 | tail -1 | cut -c1-220
tail -2 /tmp/err_$tag.txt | cut -c1-200
