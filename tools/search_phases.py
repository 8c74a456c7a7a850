"""Kernel trace (rocprofv3 --kernel-trace, csv) of a lock-step tower search -> the launches between the last step of
one replayed search and the first step of the next (result extraction, root creation and expansion), with gaps."""
import csv
import sys


def main(path, anchor="k_commit_backup_select"):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    bounds = [(idx[j], idx[j + 1]) for j in range(len(idx) - 1) if idx[j + 1] - idx[j] > 12]
    if not bounds:
        sys.exit("no search boundary found")
    a, b = bounds[-1]
    t0 = prev = int(rows[a]["End_Timestamp"])
    print(f"# launches {a + 1}..{b}: start (us), gap before (us), duration (us), kernel")
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} {(s - prev) / 1e3:7.1f} {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:96]}")
        prev = max(prev, e)
    # the whole search that follows
    nxt = [j for j in idx if j > b]
    if nxt:
        end = int(rows[nxt[min(len(nxt), 10**9) - 1]]["End_Timestamp"])
        print(f"# from the previous search's last step to the end of this trace's last step: {(end - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main(*sys.argv[1:])
