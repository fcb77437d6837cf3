set -o pipefail
# usage (GPU box): tools/trace_shard.sh -- rocprofv3 kernel trace of ONE rank of the 8-rank C4 job emulated on one GPU (bench.py --emulate-rank-of 8);
# prints the last step kernel by kernel (duration, idle gap in front) -> profiles/r03_shard8_trace.txt
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/g8
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/g8 -o r -- python3 $R/bench.py --emulate-rank-of 8 --no-cpu-baseline --no-clock-probe --steps 3 --warmup 2 > $R/gpurun_out/g8/line.json 2> $R/gpurun_out/g8/err.txt || { tail -5 $R/gpurun_out/g8/err.txt; exit 1; }
cd $R && python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/g8/**/r_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last pack of queries to the end
names=[r["Kernel_Name"] for r in rows]
votes=[i for i,n in enumerate(names) if "vote" in n]
end=votes[-1]; start=votes[-2]+1
prev=None
for r in rows[start:end+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    n=r["Kernel_Name"].split("(")[0].replace("void ","")[:50]
    gap=(s-prev)/1e3 if prev else 0
    print(f"{n:50s} wg {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):7d} {(e-s)/1e3:9.1f} us gap {gap:6.1f}")
    prev=max(prev or e,e)
print("span ms", (prev-int(rows[start]["Start_Timestamp"]))/1e6)
PY
