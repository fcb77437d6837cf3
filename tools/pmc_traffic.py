#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into per-launch bytes of mi::sim_kernel.

Units and corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports both counters in KiB; on gfx950
FETCH_SIZE tallies each 128-byte request of a wide coalesced read as 64 bytes, so it is doubled; WRITE_SIZE is
exact.  The counters sit on the L2's memory-side port: Infinity-Cache hits are included, i.e. this is L2<->fabric
traffic, an upper bound of what reaches HBM."""
import csv
import glob
import json
import sys

csv.field_size_limit(1 << 30)


def mean_of(root, counter):
    vals = []
    for path in glob.glob(f"{root}/{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            # the launches bench.py times: bootstrap + filter launches (not the speculation fallback's empty ones)
            if "sim_kernel" in r["Kernel_Name"] and "false, true>" not in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)


if __name__ == "__main__":
    root = sys.argv[1]
    f, nf = mean_of(root, "FETCH_SIZE")
    w, nw = mean_of(root, "WRITE_SIZE")
    out = {"fetch_size_kib_raw_mean": f, "write_size_kib_mean": w, "launches": nf,
           "hbm_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0}
    print(json.dumps(out))
