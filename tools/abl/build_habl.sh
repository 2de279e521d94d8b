#!/bin/bash
# Dev tool: ablation builds of the bf16-resident forward kernel (conv_h.hip, -DPG_HABL=n) linked against the current objects
# into tools/abl/libphasegen_habl<n>.so; tools/abl/habl_bench.py times U0 / U1 / D1 forward with each.
set -e
cd "$(dirname "$0")/../../unet-phasegen_amd/csrc"
make -j8 > /dev/null
for n in ${HABL:-1 2 3 4 5 6 7 8 9 10 11}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -ffp-contract=off -DPG_HABL=$n -c conv_h.hip -o build/conv_h_abl$n.o
  objs=$(ls build/*.o | grep -v conv_h | tr '\n' ' ')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/conv_h_abl$n.o -o ../../tools/abl/libphasegen_habl$n.so
done
echo built
