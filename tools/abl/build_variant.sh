#!/bin/bash
# Dev tool: a variant library with extra -D flags on every translation unit: tools/abl/build_variant.sh <name> <flags...>
# -> tools/abl/libphasegen_<name>.so (tools/abl/habl_bench.py <name> loads it).
set -e
name=$1; shift
cd "$(dirname "$0")/../../unet-phasegen_amd/csrc"
mkdir -p build/var_$name
for f in common pointwise stft conv_igemm conv_im2col conv_raw conv_raw_tall conv_raw_wgrad conv_h; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -ffp-contract=off "$@" -c $f.hip -o build/var_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build/var_$name/*.o -o ../../tools/abl/libphasegen_$name.so
echo built $name
