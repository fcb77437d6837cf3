#!/bin/bash
# usage (GPU box): tools/shard_cost.sh -- per-rank cost of the N-GPU strong-scaling bench, emulated on one GPU:
# the same 50 000 queries against a bank of 1.28M / G rows (what ONE rank of a G-rank job computes, no exchange).
set -o pipefail
for G in 1 2 4 8; do
  rows=$((1280000 / G))
  timeout -k 10 200 python bench.py --bank $rows --no-cpu-baseline "$@" > gpurun_out/shard_$G.json 2> gpurun_out/shard_$G.err || { tail -5 gpurun_out/shard_$G.err; exit 1; }
  python3 - $G <<'PY'
import json,sys
G=int(sys.argv[1]); j=json.load(open(f"gpurun_out/shard_{G}.json"))
r=j["roofline"]
print(f"G={G} ms/step={j['ms_per_step']:.2f} ideal-share={j['ms_per_step']*G:.1f} simTF={r['achieved']:.0f} sim_share={r['share_of_step_time']:.3f} sel_ms={r['select_kernel_ms_per_step']:.2f} launches/step={r['launches']/j['steps']:.0f}")
PY
done
