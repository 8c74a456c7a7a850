#!/bin/bash
# End-of-round evidence -> gpurun_out/r2f/ (copied to profiles/ afterwards): search counters and kernel stats
# (tools/r2_profiles.sh), tower search at three sizes, a self-play move with the shipped network shapes.
set -o pipefail
out=$PWD/gpurun_out/r2f; mkdir -p $out
bash tools/r2_profiles.sh || exit 1
for g in 256 1024 4096; do
  timeout -k 10 200 python tools/bench_towers.py --games $g --steps 5 > $out/towers_bench_$g.json 2>/dev/null || exit 1
done
timeout -k 10 300 python tools/bench_selfplay_towers.py > $out/selfplay_towers.json 2>/dev/null || exit 1
echo done
