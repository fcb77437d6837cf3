#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel (name prefix), mean of each counter per dispatch."""
import csv
import glob
import sys
from collections import defaultdict


def main(root, needle):
    acc = defaultdict(list)
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if needle in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        v = acc[k]
        print(f"{k},{len(v)},{sum(v) / len(v):.6g}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
