"""Turns the counter passes of tools/r2_profiles.sh (gpurun_out/r2p/{fetch,write,sq}) into the two summaries the
bench line cites: profiles/<prefix>_search_pmc_traffic.json (HBM bytes per launch of the search
kernel: FETCH_SIZE doubled as gfx950 needs for wide coalesced reads, WRITE_SIZE as read) and
profiles/<prefix>_search_pmc_sq.json (instruction and cycle counters, averaged over the launches seen).

usage: python tools/summarize_search_pmc.py [gpurun_out/r2p] [r2]"""
import collections
import csv
import glob
import json
import os
import sys


def counters(folder, kernel="k_search"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main() -> None:
    src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r2p"
    prefix = sys.argv[2] if len(sys.argv) > 2 else "r2"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fetch, write = counters(os.path.join(src, "fetch")), counters(os.path.join(src, "write"))
    f_kb, n = fetch["FETCH_SIZE"]
    w_kb, _ = write["WRITE_SIZE"]
    traffic = {"kernel": "k_search<7>", "workload": "c4_mlp, 4096 trees x 50 sims", "launches_averaged": n,
               "FETCH_SIZE_raw_KB": f_kb, "WRITE_SIZE_raw_KB": w_kb, "fetch_bytes_corrected": 2 * f_kb * 1024,
               "write_bytes": w_kb * 1024, "hbm_bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024),
               "note": "separate --pmc passes (tools/r2_profiles.sh); FETCH_SIZE doubled for gfx950 (it reports half of the "
                       "bytes of wide coalesced reads), WRITE_SIZE exact for 16-byte streaming stores; both in KB as "
                       "rocprofv3 derives them"}
    with open(os.path.join(root, "profiles", f"{prefix}_search_pmc_traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    sq = {k: v for k, (v, _) in sorted(counters(os.path.join(src, "sq")).items())}
    with open(os.path.join(root, "profiles", f"{prefix}_search_pmc_sq.json"), "w") as fh:
        json.dump(sq, fh, indent=1)
    print(json.dumps({"traffic": traffic["hbm_bytes_per_launch"], "sq": sq}))


if __name__ == "__main__":
    main()
