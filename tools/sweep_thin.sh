#!/bin/bash
# Tower search throughput against the two thresholds (one-wave kernel from N calls on, fused fc_merge + convolution
# from N rows on) at 256 / 1024 / 4096 trees -> gpurun_out/sweep_thin.txt
out=gpurun_out/sweep_thin.txt; : > $out
for g in 256 1024 4096; do
  for thin in 1 1024; do
    for fuse in 1 1024; do
      r=$(MZ_TOWER_THIN_MIN_CALLS=$thin MZ_FUSE_MIN_ROWS=$fuse timeout -k 10 120 python tools/bench_towers.py --games $g --steps 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))") || exit 1
      echo "games=$g thin_min=$thin fuse_min=$fuse sims/s,ms: $r" | tee -a $out
    done
  done
done
