#!/bin/bash
# Build an alternative libdsg (same ABI) with extra -D flags on conv_h2.hip, for A/B runs via DSG_LIB_PATH.
# Usage: tools/build_variant.sh <name> [-DFLAG ...]   ->  drivescenegen_amd/lib/libdsg_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
b=drivescenegen_amd/csrc/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c drivescenegen_amd/csrc/conv_h2.hip -o $b/conv_h2_$name.o
objs=$(ls $b/*.o | grep -v "conv_h2" )
hipcc --offload-arch=gfx950 -shared -fPIC -o drivescenegen_amd/lib/libdsg_$name.so $objs $b/conv_h2_$name.o -Wl,-rpath,/opt/rocm/lib
echo built drivescenegen_amd/lib/libdsg_$name.so
