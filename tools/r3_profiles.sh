#!/bin/bash
# Collects this round's profiling evidence on the GPU box into gpurun_out/r3p/ (summaries are copied to profiles/
# afterwards): kernel statistics of the headline bench, and FETCH_SIZE / WRITE_SIZE / SQ counter passes (one
# rocprofv3 run each) of the fused search, the residual-tower search at 256 and 4096 trees and the convolutional
# representation network.   usage: bash tools/r3_profiles.sh [search|towers|conv ...]   (default: all)
set -o pipefail
out=$PWD/gpurun_out/r3p; mkdir -p $out
root=$PWD
what=${@:-search towers conv}
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY"
cd /tmp && export TMPDIR=/tmp
passes() {  # passes <tag> <program and args...>: three counter runs of one program
  tag=$1; shift
  rm -rf $out/$tag; mkdir -p $out/$tag
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$tag/fetch -- "$@" > $out/$tag/fetch.log 2>&1 || return 1
  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$tag/write -- "$@" > $out/$tag/write.log 2>&1 || return 1
  timeout -s KILL 150 rocprofv3 --pmc $SQ --output-format csv -d $out/$tag/sq -- "$@" > $out/$tag/sq.log 2>&1 || return 1
}
for w in $what; do
  case $w in
  search)
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 1 --selfplay-moves 0 --tower-games --more-games > $out/bench_under_rocprof.json 2> $out/bench.err || exit 1
    cp $(ls -t $out/bench/*/*kernel_stats.csv | head -1) $out/bench_kernel_stats.csv
    passes search python3 $root/tools/prof_search.py || exit 1
    python3 $root/tools/summarize_pmc.py $out/search $out/search_pmc.json --only k_search --workload "c4_mlp, 4096 trees x 50 sims, one merge row per expansion (tools/prof_search.py)" || exit 1
    passes search_stored python3 $root/tools/prof_search.py --store-latents || exit 1
    python3 $root/tools/summarize_pmc.py $out/search_stored $out/search_stored_pmc.json --only k_search --workload "c4_mlp, 4096 trees x 50 sims, every node's latent stored (tools/prof_search.py --store-latents)" || exit 1
    ;;
  towers)
    for g in 256 4096; do
      passes towers_$g python3 $root/tools/bench_towers.py --games $g --steps 2 || exit 1
      python3 $root/tools/summarize_pmc.py $out/towers_$g $out/towers_${g}_pmc.json --min-launch-us 3 --workload "residual-tower search, $g trees x 50 sims, root values, no stored latents (tools/bench_towers.py --games $g --steps 2)" || exit 1
    done
    ;;
  conv)
    passes conv python3 $root/tools/bench_conv_repr.py --no-torch --iters 1 || exit 1
    python3 $root/tools/summarize_pmc.py $out/conv $out/conv_pmc.json --min-launch-us 3 --workload "shipped convolutional h, 256 games x [128,96,96] (tools/bench_conv_repr.py --no-torch --iters 1)" || exit 1
    ;;
  esac
done
echo done
