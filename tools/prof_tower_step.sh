#!/bin/bash
# Kernel trace of the residual-tower lock-step search at 256 and 4096 trees -> gpurun_out/twp/ (timeline of one step).
set -o pipefail
out=$PWD/gpurun_out/twp; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
for g in 256 4096; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t$g -- python3 $root/tools/bench_towers.py --games $g --steps 2 > $out/bench_$g.json 2> $out/err_$g.txt || exit 1
  f=$(ls -t $out/t$g/*/*kernel_trace.csv | head -1)
  python3 $root/tools/step_timeline.py $f ${MZ_TIMELINE_ANCHOR:-k_commit_backup_select} > $out/timeline_$g.txt || exit 1
  cp $(ls -t $out/t$g/*/*kernel_stats.csv | head -1) $out/kernel_stats_$g.csv
done
echo done
