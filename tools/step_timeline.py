"""One simulation step of a lock-step search as a kernel timeline, from a rocprofv3 --kernel-trace csv.

usage: python tools/step_timeline.py <kernel_trace.csv> [anchor kernel substring, default k_backup]

Takes the last complete span between two launches of the anchor kernel (one simulation step: the step's launches
replayed from the HIP graph) and prints start (us, from the span's first launch), duration (us), queue, grid,
workgroup and kernel name; then the per-kernel sums over all such spans."""
import csv
import sys
from collections import defaultdict


def main() -> None:
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_backup"
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"),
                         r["Workgroup_Size_X"] if "Workgroup_Size_X" in r else r.get("Workgroup_Size", "?"), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if anchor in r[5]]
    if len(marks) < 3:
        raise SystemExit(f"fewer than three launches of {anchor}")
    a, b = marks[-3] + 1, marks[-2] + 1  # the last span but one: the search's last step may differ
    t0 = rows[a][0]
    queues = {}
    print(f"# one step: launches {a}..{b - 1} of {len(rows)}; span {(rows[b - 1][1] - t0) / 1e3:.1f} us")
    for s, e, q, grid, wg, name in rows[a:b]:
        queues.setdefault(q, f"q{len(queues) + 1}")
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {queues[q]} grid={grid} wg={wg} {name[:70]}")
    total = defaultdict(lambda: [0, 0.0])
    spans = 0
    for m0, m1 in zip(marks[:-1], marks[1:]):
        if m1 - m0 != b - a:
            continue
        spans += 1
        for s, e, _, _, _, name in rows[m0 + 1:m1 + 1]:
            total[name[:70]][0] += 1
            total[name[:70]][1] += (e - s) / 1e3
    print(f"# per step, averaged over {spans} steps of the same launch count: launches, summed duration (us)")
    for name, (n, us) in sorted(total.items(), key=lambda kv: -kv[1][1]):
        print(f"{n / spans:6.1f} {us / spans:8.1f} {name}")


if __name__ == "__main__":
    main()
