#!/bin/bash
# SQ counters of the fused search kernel (tools/prof_search.py), one rocprofv3 pass.
# usage: tools/pmc_search.sh <outdir> <tag> [counters...]
out=$1; tag=$2; shift 2
ctrs=${@:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY}
mkdir -p "$out"; rm -rf "$out/pmc_$tag"
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pmc_$tag" -- python3 tools/prof_search.py > "$out/log_$tag.txt" 2>&1 || { echo "$tag failed"; exit 1; }
python3 - "$out" "$tag" <<'PY'
import sys, glob, csv, collections
out, m = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list); dur = []
for f in glob.glob(f"{out}/pmc_{m}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_search" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta = (r["Kernel_Name"][:40], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"])
line = {k: sum(v) / len(v) for k, v in acc.items()}
line["us"] = sum(dur) / max(len(dur), 1) / 1e3 / max(len(acc), 1)* len(acc)
print(m, meta, " ".join(f"{k}={v:.4g}" for k, v in sorted(line.items())))
PY
