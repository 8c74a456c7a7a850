"""Per-kernel summary of rocprofv3 counter passes.

    python tools/summarize_pmc.py <dir> <out.json> [--workload TEXT] [--only SUBSTR ...]

<dir> holds one sub-folder per pass (`fetch`: FETCH_SIZE, `write`: WRITE_SIZE, `sq`: SQ_* counters; any may be
missing), each the -d folder of one `rocprofv3 --pmc ... --output-format csv` run of the same program.  Per kernel
name: launches seen, mean counter values per launch, HBM bytes per launch (FETCH_SIZE doubled, as gfx950 needs for
wide coalesced reads: it tallies 128-byte requests at 64 bytes; WRITE_SIZE as read; both arrive in KB), executed
matrix flops when SQ_INSTS_MFMA is there (given --mfma-flop per instruction), matrix-pipe busy share
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)."""
import argparse
import collections
import csv
import glob
import json
import os
import re


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = re.sub(r"\s*\[clone.*$", "", name)
    return name[:90]


def counters(folder, only):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if only and not any(o in k for o in only):
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "End_Timestamp" in r:
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc, dur


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("out")
    ap.add_argument("--workload", default="")
    ap.add_argument("--only", nargs="*", default=[])
    ap.add_argument("--mfma-flop", type=float, default=16384.0,
                    help="flops per matrix instruction counted (16x16x32 f16/bf16: 16384; 16x16x4 f32: 2048), per wave")
    ap.add_argument("--min-launch-us", type=float, default=0.0)
    args = ap.parse_args()
    out = collections.defaultdict(dict)
    for leg in ("fetch", "write", "sq"):
        folder = os.path.join(args.src, leg)
        if not os.path.isdir(folder):
            continue
        acc, dur = counters(folder, args.only)
        for k, d in acc.items():
            e = out[k]
            for c, v in d.items():
                e[c] = sum(v) / len(v)
                e["launches_" + leg] = len(v)
            if dur[k]:
                e["avg_us_under_" + leg] = sum(dur[k]) / len(dur[k]) / 1e3
    for k, e in out.items():
        if "FETCH_SIZE" in e:
            e["fetch_bytes_corrected"] = 2 * e["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in e:
            e["write_bytes"] = e["WRITE_SIZE"] * 1024
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = int(e["fetch_bytes_corrected"] + e["write_bytes"])
        if "SQ_INSTS_MFMA" in e:
            e["executed_matrix_flop_per_launch"] = e["SQ_INSTS_MFMA"] * args.mfma_flop
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CU_CYCLES"):
            e["matrix_pipe_busy_share"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * e["SQ_BUSY_CU_CYCLES"])
    keep = {k: e for k, e in sorted(out.items()) if max([v for n, v in e.items() if n.startswith("avg_us")] or [1e9]) >= args.min_launch_us}
    doc = {"workload": args.workload,
           "note": "separate --pmc passes of the same program; FETCH_SIZE / WRITE_SIZE arrive in KB; FETCH_SIZE doubled "
                   "(gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes), WRITE_SIZE as read; "
                   "durations are under counter collection (serialised launches), not the benchmarked ones",
           "kernels": keep}
    with open(args.out, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, e in keep.items():
        print(k, {n: (round(v, 3) if isinstance(v, float) else v) for n, v in e.items()
                  if n in ("hbm_bytes_per_launch", "matrix_pipe_busy_share", "executed_matrix_flop_per_launch", "launches_sq", "launches_fetch")})


if __name__ == "__main__":
    main()
