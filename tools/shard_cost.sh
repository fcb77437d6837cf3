#!/bin/bash
# usage (GPU box): tools/shard_cost.sh [bench args] -- per-rank cost of the N-GPU strong-scaling bench, emulated on one GPU:
# what ONE rank of a G-rank job computes -- the same 50 000 queries against its 1.28M / G bank rows, then merge + vote for
# its 1 / G slice of the queries -- without the exchange (bench.py --emulate-rank-of G).  G x time(G) / time(1) bounds the
# scaling efficiency from above.  Writes gpurun_out/shard_cost.jsonl (one line per G) besides the table.
set -o pipefail
: > gpurun_out/shard_cost.jsonl
for G in 1 2 4 8; do
  E=""; [ $G -gt 1 ] && E="--emulate-rank-of $G"
  timeout -k 10 200 python bench.py $E --no-cpu-baseline --no-clock-probe "$@" > gpurun_out/shard_$G.json 2> gpurun_out/shard_$G.err || { tail -5 gpurun_out/shard_$G.err; exit 1; }
  python3 - $G <<'PY'
import json,sys
G=int(sys.argv[1]); j=json.load(open(f"gpurun_out/shard_{G}.json"))
r=j["roofline"]
base=json.load(open("gpurun_out/shard_1.json"))["ms_per_step"]
row={"G":G,"ms_per_step":round(j["ms_per_step"],2),"efficiency_upper_bound":round(base/(G*j["ms_per_step"]),3),"sim_tflops":round(r["achieved"]),
     "sim_share":round(r["share_of_step_time"],3),"refresh_ms":round(r["select_kernel_ms_per_step"],2),"sim_launches_per_step":round(r["launches"]/j["steps"])}
open("gpurun_out/shard_cost.jsonl","a").write(json.dumps(row)+"\n")
print(row)
PY
done
