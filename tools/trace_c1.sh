#!/bin/bash
# usage (GPU box): tools/trace_c1.sh <tag> <fp32|bf16> -- rocprofv3 kernel trace of tools/exp_c1.py (C1: 10k x 50k, D = 128); prints the
# LAST knn_predict's kernels in launch order with duration and the idle gap in front of each -> gpurun_out/<tag>_trace.txt
set -o pipefail
T=$1; DT=${2:-bf16}; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$T -o r -- python3 $R/tools/exp_c1.py $DT > $R/gpurun_out/trace_$T.out 2> $R/gpurun_out/trace_$T.err || { tail -5 $R/gpurun_out/trace_$T.err; exit 1; }
cat $R/gpurun_out/trace_$T.out
cd $R && python3 - $T <<'PY'
import csv, glob, sys
t = sys.argv[1]
f = glob.glob(f"gpurun_out/trace_{t}/**/r_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "pack_kernel" in n or "pack_rows" in n]    # a call = bank pack, query pack, ...
votes = [i for i, n in enumerate(names) if "vote" in n]
last_vote = votes[-1]
first = max(i for i in starts if i < last_vote and ("pack_kernel" in names[i]))          # the transposed-bank pack opens the call
out, prev_end, tot = [], None, {}
for r in rows[first:last_vote + 1]:
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
