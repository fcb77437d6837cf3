#!/bin/bash
# usage (GPU box): tools/ab_env.sh "<VAR=val> ..." [rounds] [bench args]  -- interleaved A/B of environment settings on ONE box
R=${2:-2}
ARGS=${3:---queries 16384 --query-chunk 16384}
for r in $(seq 1 $R); do for s in $1; do
  env $s timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/ab.json')); r=d['roofline']; print('$s', round(d['value']), round(d['ms_per_step'],1), 'simTF', round(r['achieved']), 'sel', round(r['select_kernel_ms_per_step'],1))"; done; done
