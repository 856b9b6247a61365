#!/bin/bash
# developer tool: builds two experiment libraries next to the product library
#   libvolviz_x_NOLOAD.so  march_kernel with the volume gathers replaced by register data (VALU side only)
#   libvolviz_x_NOALU.so   march_kernel with classification/blending replaced by a sum (gather side only)
# and leaves the product build untouched.  Run on the GPU with
#   VV_STATS=1 VV_LIB=$PWD/volume-viz_amd/lib/libvolviz_x_NOLOAD.so python bench.py --no-cpu-baseline --size 64
# (stats[1] = lane slots spent; DESIGN.md section 4 has the resulting decomposition).
set -e
cd "$(dirname "$0")/../volume-viz_amd"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value"
for X in NOLOAD NOALU; do
  mkdir -p build_x
  for f in vv_raymarch vv_raymarch_big vv_raymarch_brick vv_raymarch_zpair vv_raymarch_wstaged vv_aux; do /opt/rocm/bin/hipcc $FLAGS -DVV_X_$X -c csrc/$f.hip -o build_x/$f.o & done
  /opt/rocm/bin/hipcc $FLAGS -DVV_X_$X -x hip -c csrc/vv_api.cpp -o build_x/vv_api.o &
  g++ -O2 -std=c++17 -fPIC -c csrc/vv_host.cpp -o build_x/vv_host.o &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libvolviz_x_$X.so build_x/*.o
  rm -rf build_x
done
ls -la lib
