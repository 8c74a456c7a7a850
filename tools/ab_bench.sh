#!/bin/bash
# A/B of two builds of the library on one GPU box: alternates them so that box-to-box and
# thermal differences cancel.  usage: tools/ab_bench.sh libA.so libB.so
for i in 1 2; do
  for lib in "$@"; do
    echo -n "$lib: "
    MUZERO_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tower-games --selfplay-moves 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M sims/s', round(d['ms_per_step'],3), 'ms')" || exit 1
  done
done
