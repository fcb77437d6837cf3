#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: for kernels whose name contains <needle>, restricted to the
dispatches of the grid size with the largest mean counter values (the full-size filter stages), the mean of each counter per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def main(root, needle):
    rows = []
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if needle in r["Kernel_Name"] and "false, true>" not in r["Kernel_Name"]:      # not the (empty) fallback launches
                rows.append((int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
    if not rows:
        print("no dispatches matched")
        return
    # the grid that did the work: largest mean counter value per dispatch (an empty fallback launch can have the larger grid)
    tot, cnt = defaultdict(float), defaultdict(int)
    for g, _, v in rows:
        tot[g] += v
        cnt[g] += 1
    big = max(tot, key=lambda g: tot[g] / cnt[g])
    acc = defaultdict(list)
    for g, n, v in rows:
        if g == big:
            acc[n].append(v)
    print(f"# grid {big} ({big // 512} workgroups), per-dispatch means")
    for k in sorted(acc):
        v = acc[k]
        print(f"{k},{len(v)},{sum(v) / len(v):.6g}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
