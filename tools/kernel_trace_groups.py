"""Per-dispatch durations of a rocprofv3 --kernel-trace CSV, grouped by kernel and grid size.

    python tools/kernel_trace_groups.py <kernel_trace.csv> <substring of the kernel name>
"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    pat = sys.argv[2]
    agg = collections.defaultdict(list)
    for r in rows:
        if pat in r["Kernel_Name"]:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            agg[(name[:48], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["LDS_Block_Size"]))].append(d)
    for k, v in sorted(agg.items()):
        v.sort()
        print(f"{k[0]:50s} wgs {k[1]:6d} lds {k[2]:7d} calls {len(v):5d}  min {v[0]:7d}  median {v[len(v) // 2]:7d}  max {v[-1]:7d} ns")


if __name__ == "__main__":
    main()
