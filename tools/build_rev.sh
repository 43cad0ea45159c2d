#!/bin/bash
# Build libdsg of a git revision into drivescenegen_amd/lib/libdsg_<name>.so (same ABI assumed) for A/B runs on ONE box:
#   tools/build_rev.sh base HEAD   ->  DSG_LIB_PATH=drivescenegen_amd/lib/libdsg_base.so python tools/fwd_bench.py ...
# (boxes differ by several percent among themselves; only numbers taken inside one gpurun call compare)
set -e
cd "$(dirname "$0")/.."
name=$1; rev=${2:-HEAD}
wt=/tmp/dsg_rev_$name
rm -rf $wt; git worktree prune
git worktree add -f --detach $wt $rev > /dev/null
python $wt/drivescenegen_amd/csrc/build.py | tail -1
cp $wt/drivescenegen_amd/lib/libdsg.so drivescenegen_amd/lib/libdsg_$name.so
git worktree remove --force $wt
echo built drivescenegen_amd/lib/libdsg_$name.so from $rev
