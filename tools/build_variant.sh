#!/bin/bash
# Build an alternative libdsg (same ABI) with extra -D flags on ONE source file, for A/B runs via DSG_LIB_PATH.
# Usage: tools/build_variant.sh <name> <source.hip> [-DFLAG ...]   ->  drivescenegen_amd/lib/libdsg_<name>.so
#        (source defaults to conv_h2.hip when the second argument starts with -D)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=conv_h2.hip
case "$1" in -D*) ;; *) src=$1; shift;; esac
stem=${src%.hip}
b=drivescenegen_amd/csrc/build; mkdir -p tools/_build
extra=""
[ "$src" = "scheduler.hip" ] && extra="-ffp-contract=off"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $extra "$@" -c drivescenegen_amd/csrc/$src -o tools/_build/variant_${stem}_$name.obj
objs=$(ls $b/*.o | grep -v "/${stem}\.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o drivescenegen_amd/lib/libdsg_$name.so $objs tools/_build/variant_${stem}_$name.obj -Wl,-rpath,/opt/rocm/lib
echo built drivescenegen_amd/lib/libdsg_$name.so
