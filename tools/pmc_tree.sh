#!/bin/bash
# HBM traffic of the lock-step tree kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
# over tools/bench_tree.py --games 524288 --sims 10; mean of the last 10 launches of each kernel.
# usage: tools/pmc_tree.sh <outdir> <result.json>
out=$1; res=$2
mkdir -p "$out"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$out/pmc_tree_$c"
  timeout -s KILL 170 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_tree_$c" -- python3 tools/bench_tree.py --games 524288 --sims 10 > "$out/log_tree_$c.txt" 2>&1 || { echo "$c pass failed"; exit 1; }
done
python3 - "$out" "$res" <<'PY'
import sys, glob, csv, collections, json
out, res = sys.argv[1], sys.argv[2]
G = 524288
line = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/pmc_tree_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for k in ("k_select", "k_expand_commit", "k_backup"):
                if k in r["Kernel_Name"] and r["Counter_Name"] == c:
                    acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[-10:]
        line.setdefault(k, {})[c + "_KB_raw"] = sum(v) / len(v)
for k, d in line.items():
    # gfx950: FETCH_SIZE tallies 128-byte requests as 64 bytes, so reads are between 1x and 2x the raw figure
    d["fetch_bytes_per_tree_raw"] = d["FETCH_SIZE_KB_raw"] * 1024 / G
    d["fetch_bytes_per_tree_upper"] = 2 * d["fetch_bytes_per_tree_raw"]
    d["write_bytes_per_tree"] = d["WRITE_SIZE_KB_raw"] * 1024 / G
line["workload"] = "tools/bench_tree.py --games 524288 --sims 10, mean of the last 10 launches of each kernel (rocprofv3 --pmc, separate passes)"
json.dump(line, open(res, "w"), indent=1)
print(json.dumps(line))
PY
