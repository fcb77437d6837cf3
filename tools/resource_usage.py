#!/usr/bin/env python3
"""Per-kernel resource table of libmi_knn.so from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python tools/resource_usage.py [> profiles/rNN_kernel_resources.txt]

Rebuilds the library (cross-compiles: no GPU needed) and prints, per kernel: VGPRs, AGPRs, scratch bytes per lane,
LDS bytes and waves per SIMD.  Exits non-zero if any kernel uses scratch (a register spill in a product kernel)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "image-classification_amd", "_build.py")],
                         capture_output=True, text=True, check=True)
    text = out.stdout + out.stderr
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]|SGPRs): (\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"name": subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip() or val}
            rows.append(cur)
        elif cur is not None:
            cur[key.split(" ")[0]] = int(val)
    print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS':>7} {'occ':>4}  kernel")
    bad = 0
    for r in rows:
        name = re.sub(r"\((?!anonymous).*", "", r["name"]).replace("(anonymous namespace)::", "")
        print(f"{r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('SGPRs', 0):5d} {r.get('ScratchSize', 0):8d} "
              f"{r.get('LDS', 0):7d} {r.get('Occupancy', 0):4d}  {name}")
        bad += r.get("ScratchSize", 0) > 0
    print(f"{len(rows)} kernels, {bad} with scratch")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
