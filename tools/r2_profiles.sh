#!/bin/bash
# Collects the round's profiling evidence on the GPU box into gpurun_out/r2p/ (copied to profiles/ afterwards).
set -o pipefail
out=$PWD/gpurun_out/r2p; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/tools/prof_search.py > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/tools/prof_search.py > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/sq -- python3 $root/tools/prof_search.py > /dev/null 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/conv -- python3 $root/tools/bench_conv_repr.py --no-torch --iters 3 > $out/conv_under_rocprof.json 2>/dev/null || exit 1
cd $root
timeout -k 10 200 python tools/bench_conv_repr.py > $out/conv_bench.json 2>/dev/null || exit 1
timeout -k 10 200 python tools/bench_conv_repr.py --fp32 --no-torch > $out/conv_bench_fp32.json 2>/dev/null || exit 1
echo done
