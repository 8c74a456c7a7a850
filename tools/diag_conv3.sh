#!/bin/bash
# What bounds k_conv3_same: builds of the library with parts of the kernel left out (-DMZ_CONV3_SKIP=mask,
# see csrc/mz_conv3.hip), each timed by tools/bench_conv3_kernel.py.  Build the variants first, on the CPU:
#   tools/diag_conv3.sh build 1 2 4 6 8 16 ...      (writes muzero_amd/_diag/libmuzero_hip_c3_<mask>.so)
# then on the GPU:  tools/diag_conv3.sh run <outfile> 0 1 2 4 6 8 16 ...   (mask 0 = the shipped library)
set -e
mode=$1; shift
cd "$(dirname "$0")/.."
if [ "$mode" = build ]; then
  mkdir -p muzero_amd/_diag
  for m in "$@"; do
    hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Iinclude -Imuzero_amd/csrc -DMZ_CONV3_SKIP=$m -c muzero_amd/csrc/mz_conv3.hip -o muzero_amd/_diag/mz_conv3_$m.o
    objs=$(ls muzero_amd/csrc/_obj/*.o | grep -v "mz_conv3.o\|_diag")
    hipcc -shared -fPIC --offload-arch=gfx950 -o muzero_amd/_diag/libmuzero_hip_c3_$m.so $objs muzero_amd/_diag/mz_conv3_$m.o
    rm muzero_amd/_diag/mz_conv3_$m.o
  done
else
  out=$1; shift
  : > "$out"
  for m in "$@"; do
    if [ "$m" = 0 ]; then unset MUZERO_HIP_LIB; else export MUZERO_HIP_LIB=$PWD/muzero_amd/_diag/libmuzero_hip_c3_$m.so; fi
    timeout -k 10 120 python3 tools/bench_conv3_kernel.py >> "$out" || { echo "mask $m failed"; exit 1; }
  done
fi
