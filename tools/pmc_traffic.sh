#!/bin/bash
# usage (GPU box): tools/pmc_traffic.sh <tag>  -- FETCH_SIZE and WRITE_SIZE of the default bench.py workload, one pass each
# (MI355X_MICROARCH.md: separate --pmc passes; FETCH_SIZE is in KiB-units of 64-B-tallied 128-B requests on gfx950: x2)
set -o pipefail
T=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/traffic_$T
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 180 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/traffic_$T/$c -o r -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/traffic_$T/$c.json 2> $R/gpurun_out/traffic_$T/$c.log || { tail -5 $R/gpurun_out/traffic_$T/$c.log; exit 1; }
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/traffic_$T
