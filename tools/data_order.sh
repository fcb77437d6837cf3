#!/bin/bash
# usage (GPU box): tools/data_order.sh [bench args] -- C4 on random / clustered / class-sorted clustered data, one box -> gpurun_out/data_order.jsonl
set -o pipefail
: > gpurun_out/data_order.jsonl
for d in random clustered clustered-sorted; do
  timeout -k 10 300 python bench.py --data $d --no-cpu-baseline --no-clock-probe "$@" > gpurun_out/data_$d.json 2> gpurun_out/data_$d.err || { tail -5 gpurun_out/data_$d.err; exit 1; }
  python3 - $d "$*" <<'PY'
import json,sys
d=sys.argv[1]; j=json.load(open(f"gpurun_out/data_{d}.json")); r=j["roofline"]
row={"data":d,"args":sys.argv[2],"queries_per_s":round(j["value"]),"ms_per_step":round(j["ms_per_step"],2),"sim_tflops":round(r["achieved"]),"refresh_ms":round(r["select_kernel_ms_per_step"],2)}
open("gpurun_out/data_order.jsonl","a").write(json.dumps(row)+"\n"); print(row)
PY
done
