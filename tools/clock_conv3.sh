#!/bin/bash
# Effective shader clock of the convolution kernel (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / duration),
# for the shipped library and diagnostic variants (tools/diag_conv3.sh build ...).
# usage: tools/clock_conv3.sh <outdir> <masks...>
out=$1; shift
mkdir -p "$out"
for m in "$@"; do
  if [ "$m" = 0 ]; then unset MUZERO_HIP_LIB; else export MUZERO_HIP_LIB=$PWD/muzero_amd/_diag/libmuzero_hip_c3_$m.so; fi
  rm -rf "$out/clk_$m"
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/clk_$m" -- python3 tools/bench_conv3_kernel.py --iters 200 > "$out/clk_$m.log" 2>&1 || { echo "mask $m failed"; exit 1; }
  python3 - "$out/clk_$m" "$m" <<'PY'
import sys, glob, csv
d, m = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_conv3" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            rows.append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows = rows[len(rows) // 2:]  # the later launches: the clock has settled
cyc = sum(c for c, _ in rows) / len(rows) / 8
ns = sum(t for _, t in rows) / len(rows)
print(f"mask {m}: {ns / 1e3:.1f} us, {cyc / ns:.3f} GHz effective, {len(rows)} launches")
PY
done
