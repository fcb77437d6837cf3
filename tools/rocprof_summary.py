#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short, committed summary (names truncated)."""
import csv
import sys


def main(src, dst, note=""):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        if note:
            f.write(f"# {note}\n")
        f.write("kernel,calls,total_ms,avg_us,percent,min_us,max_us\n")
        for r in rows:
            name = r["Name"].split("(")[0].replace("void ", "")[:70]
            f.write(f"{name},{r['Calls']},{int(r['TotalDurationNs']) / 1e6:.3f},{float(r['AverageNs']) / 1e3:.2f},"
                    f"{float(r['Percentage']):.3f},{int(r['MinNs']) / 1e3:.2f},{int(r['MaxNs']) / 1e3:.2f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
