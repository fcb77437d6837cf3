#!/usr/bin/env python3
"""Condense a rocprofv3 rocpd database (`*_results.db`, the default output of --kernel-trace --stats on
ROCm 7.2) into the short per-kernel CSV committed under profiles/ (names truncated)."""
import sqlite3
import sys
from collections import defaultdict


def main(src, dst, note=""):
    con = sqlite3.connect(src)
    acc = defaultdict(list)
    for name, dur in con.execute("select name, duration from kernels"):
        acc[name.split("(")[0].replace("void ", "")[:70]].append(dur)
    total = sum(sum(v) for v in acc.values())
    with open(dst, "w") as f:
        if note:
            f.write(f"# {note}\n")
        f.write("kernel,calls,total_ms,avg_us,percent,min_us,max_us\n")
        for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"{name},{len(v)},{sum(v) / 1e6:.3f},{sum(v) / len(v) / 1e3:.2f},{100.0 * sum(v) / total:.3f},"
                    f"{min(v) / 1e3:.2f},{max(v) / 1e3:.2f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
