#!/bin/bash
# Developer tool (CPU, no GPU): the CPU test suite with the oracle and the host-side C++ of the product (vv_host.cpp: .t3d
# reader / writer, camera controls, slice matrix, table presets) built with AddressSanitizer + UBSan.  (GPU sanitizers are
# not available on the test pool.)  Restores the normal libraries afterwards.  usage: bash tools/asan_cpu.sh
set -e
cd "$(dirname "$0")/.."
make -C oracle >/dev/null; make -C volume-viz_amd >/dev/null
cp oracle/_build/libvvoracle.so /tmp/libvvoracle_orig.so
cp volume-viz_amd/lib/libvolviz_hip.so /tmp/libvolviz_hip_orig.so
trap 'cp /tmp/libvvoracle_orig.so oracle/_build/libvvoracle.so; cp /tmp/libvolviz_hip_orig.so volume-viz_amd/lib/libvolviz_hip.so' EXIT
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g"
gcc $SAN -std=gnu11 -fPIC -shared -ffp-contract=off -fno-fast-math -fopenmp -o oracle/_build/libvvoracle.so oracle/vvo.c -lm
g++ $SAN -std=c++17 -fPIC -Wall -c volume-viz_amd/csrc/vv_host.cpp -o /tmp/vv_host_asan.o
B=volume-viz_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o volume-viz_amd/lib/libvolviz_hip.so $B/vv_raymarch.o $B/vv_raymarch_big.o $B/vv_raymarch_brick.o \
    $B/vv_raymarch_zpair.o $B/vv_raymarch_wstaged.o $B/vv_sweep.o $B/vv_aux.o $B/vv_api.o /tmp/vv_host_asan.o
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" python -m pytest tests/ -x -q -m "not gpu"
