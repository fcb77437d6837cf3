#!/bin/bash
# usage: tools/ab.sh "<lib-suffix:G> ..." [rounds]  -- interleaved A/B of library variants on ONE box
R=${2:-2}
for r in $(seq 1 $R); do for v in $1; do s=${v%%:*}; g=${v##*:}; lib=$GRAFT_REPO_ROOT/image-classification_amd/libmi_knn$s.so
  MI_KNN_GTILES=$g MI_KNN_LIB=$lib timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --query-chunk 16384 > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/ab.json')); r=d['roofline']; print('$v', round(d['value']), round(d['ms_per_step'],1), round(r['achieved']), round(r['select_kernel_ms_per_step'],1))"; done; done
