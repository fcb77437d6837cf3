#!/bin/bash
# usage (GPU box): tools/chunk_sweep.sh "<bank rows>" "<query chunks>" -- one box, so the lines are comparable
set -o pipefail
for rows in $1; do for qc in $2; do
  timeout -k 10 200 python bench.py --bank $rows --query-chunk $qc $3 --no-cpu-baseline > gpurun_out/cs.json 2> gpurun_out/cs.err || { tail -5 gpurun_out/cs.err; exit 1; }
  python3 - $rows $qc $3 <<'PY'
import json,sys
j=json.load(open("gpurun_out/cs.json")); r=j["roofline"]
print(f"rows={sys.argv[1]} chunk={sys.argv[2]} {' '.join(sys.argv[3:])} ms/step={j['ms_per_step']:.2f} q/s={j['value']:.0f} simTF={r['achieved']:.0f} sim_share={r['share_of_step_time']:.3f} sel_ms={r['select_kernel_ms_per_step']:.2f} launches/step={r['launches']/j['steps']:.0f}")
PY
done; done
