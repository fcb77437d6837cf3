#!/bin/bash
# usage (GPU box): tools/dim_sweep.sh -- sim kernel rate vs K (per-tile fixed overhead = what a longer K loop amortises)
set -o pipefail
for d in 1024 2048 4096 8192; do
  rows=$((1280000 * 2048 / d / 2))
  timeout -k 10 200 python bench.py --dim $d --bank $rows --queries 16384 --no-cpu-baseline > gpurun_out/ds.json 2> gpurun_out/ds.err || { tail -5 gpurun_out/ds.err; exit 1; }
  python3 -c "
import json; j=json.load(open('gpurun_out/ds.json')); r=j['roofline']
print('D=$d rows=$rows simTF', round(r['achieved']), 'avg_launch_ms', round(r['avg_launch_ms'],3), 'launches', r['launches'])"
done
