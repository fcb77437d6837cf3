#!/bin/bash
# usage (GPU box): tools/trace_step.sh <tag> [bench args] -- rocprofv3 kernel trace of bench.py (1 warm-up + 1 timed step); prints the
# timed step's kernels in launch order with duration and the idle gap in front of each, and writes gpurun_out/<tag>_trace.txt
set -o pipefail
T=$1; shift; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$T -o r -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-clock-probe "$@" > $R/gpurun_out/trace_$T.json 2> $R/gpurun_out/trace_$T.err || { tail -5 $R/gpurun_out/trace_$T.err; exit 1; }
cd $R && python3 - $T <<'PY'
import csv, glob, sys
t = sys.argv[1]
f = glob.glob(f"gpurun_out/trace_{t}/**/r_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
dense = [i for i, n in enumerate(names) if "sim_kernel<" in n and ", 0, " in n]      # bootstrap launches: the second half are the timed step's
first = dense[len(dense) // 2]
while first > 0 and any(t in names[first - 1] for t in ("pack", "init_state", "elementwise", "normalized")):
    first -= 1
out = []
prev_end = None
tot = {}
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    out.append(f"{n:48s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):7d} wg  {(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us")
    tot[n] = tot.get(n, 0.0) + (e - s) / 1e3
    tot["(idle gaps)"] = tot.get("(idle gaps)", 0.0) + max(gap, 0.0)
    prev_end = max(prev_end or e, e)
span = (prev_end - int(rows[first]["Start_Timestamp"])) / 1e3
out.append(f"span {span / 1e3:.3f} ms")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    out.append(f"  {n:48s} {v / 1e3:8.3f} ms  {100 * v / span:5.1f} %")
open(f"gpurun_out/{t}_trace.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
