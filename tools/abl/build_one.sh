#!/bin/bash
# Dev tool: a variant library that differs from the product build in ONE translation unit compiled with extra -D flags:
#   tools/abl/build_one.sh <name> <unit> <flags...>   ->   tools/abl/libphasegen_<name>.so   (needs an up-to-date csrc/build/)
set -e
name=$1; unit=$2; shift; shift
cd "$(dirname "$0")/../../unet-phasegen_amd/csrc"
mkdir -p build/one_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -ffp-contract=off "$@" -c $unit.hip -o build/one_$name/$unit.o
objs=$(ls build/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/one_$name/$unit.o -o ../../tools/abl/libphasegen_$name.so
echo built $name
