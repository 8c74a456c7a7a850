#!/bin/bash
# Diagnostic: instruction counters and duration of the fused search with stages left out
# (diagnostic library built with -DMZ_SEARCH_DIAG; bits of MZ_SEARCH_SKIP, see csrc/mz_search.hip).
# usage: tools/diag_search_pmc.sh <outdir> <skip masks...>
out=$1; shift
mkdir -p "$out"
export MUZERO_HIP_LIB=$PWD/muzero_amd/libmuzero_hip_diag.so
for m in "$@"; do
  export MZ_SEARCH_SKIP=$m
  rm -rf "$out/pmc_$m"
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
      --output-format csv -d "$out/pmc_$m" -- python3 tools/prof_search.py > "$out/log_$m.txt" 2>&1 || { echo "mask $m failed"; exit 1; }
  python3 - "$out" "$m" <<'PY'
import sys, glob, csv, collections
out, m = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
dur = []
for f in glob.glob(f"{out}/pmc_{m}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_search" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
line = {k: sum(v) / len(v) for k, v in acc.items()}
line["us"] = sum(dur) / max(len(dur), 1) / 1e3
print(m, " ".join(f"{k}={v:.4g}" for k, v in sorted(line.items())))
PY
done
