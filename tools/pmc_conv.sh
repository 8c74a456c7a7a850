#!/bin/bash
# SQ counters of the convolution kernels of the representation network (tools/bench_conv_repr.py), one
# rocprofv3 pass per counter group.  usage: tools/pmc_conv.sh <outdir> <tag> <counters...>
out=$1; tag=$2; shift 2
mkdir -p "$out"; rm -rf "$out/pmc_$tag"
timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$out/pmc_$tag" -- python3 tools/bench_conv_repr.py --no-torch --iters 1 > "$out/log_$tag.txt" 2>&1 || { echo "$tag failed"; tail -5 "$out/log_$tag.txt"; exit 1; }
python3 - "$out" "$tag" <<'PY'
import sys, glob, csv, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_conv3" in k:
            name = "k_conv3_same" if "k_conv3_same" in k else "k_conv3"
            acc[(name, r["Grid_Size"] if "Grid_Size" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(tag, k, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())), "launches", len(next(iter(d.values()))))
PY
