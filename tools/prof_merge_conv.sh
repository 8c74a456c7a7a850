#!/bin/bash
# mz_merge_conv_forward alone at 4096 trees, whole and with parts left out (MZ_MC_SKIP bits: 1 merge product, 2 convolution
# product, 8 loads of the bottleneck rows) -> gpurun_out/twp/merge_conv_parts.txt
set -o pipefail
out=$PWD/gpurun_out/twp; mkdir -p $out
: > $out/merge_conv_parts.txt
for m in 0 1 2 3 11; do
  echo "MZ_MC_SKIP=$m" >> $out/merge_conv_parts.txt
  MZ_MC_SKIP=$m timeout -k 10 100 python tools/bench_merge_conv.py 4096 2>/dev/null >> $out/merge_conv_parts.txt || exit 1
done
